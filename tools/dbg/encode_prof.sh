#!/bin/bash
# GPU tuning aid: kernel times of the batch encoder (direct and through the word cache) on the abcd and Zipf sentence sets of bench.py.
# usage (GPU box): bash tools/dbg/encode_prof.sh  -> prints the k5* rows
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/encp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/encp -- python $R/bench.py --steps 2 --warmup 0 --no-e2e --no-cpu-baseline --no-touched-pass > /tmp/encp.json 2>/tmp/encp.err
python $R/tools/pmc_summary.py kernel-stats /tmp/encp /tmp/encp.csv
grep "k5\|fill_u64\|^kernel" /tmp/encp.csv | head -20
python - <<'PY'
import json
d = json.load(open("/tmp/encp.json"))
def find(o, path=""):
    if isinstance(o, dict):
        for k, v in o.items():
            if k == "word_cache":
                print(path, o.get("value"), o.get("kernel_ms"), json.dumps(v))
            find(v, path + "/" + k)
find(d)
print(d["parity"])
PY
