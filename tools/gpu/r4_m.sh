#!/bin/bash
# kernel times of the cached encode by variant of tools/dbg/encode_ab.py (rocprofv3 --kernel-trace --stats, one variant per run)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-2 3}; do
  rm -rf /tmp/encp$v
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/encp$v -- python $R/tools/dbg/encode_ab.py 10000000 ${KIND:-abcd} $v 1 > $R/gpurun_out/m_$v.log 2>&1
  python $R/tools/pmc_summary.py kernel-stats /tmp/encp$v $R/gpurun_out/m_stats_$v.csv
  grep -E "^(abcd|zipf)" $R/gpurun_out/m_$v.log
  grep -E "k5|fill|^kernel|scan_" $R/gpurun_out/m_stats_$v.csv | head -16
done
