#!/bin/bash
# Regenerates the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun from the repo root):
#   kernel-trace --stats of one default bench step, FETCH_SIZE and WRITE_SIZE in separate --pmc passes (counters never
#   share a run with trace domains other than the kernel trace), and the bench line itself.
# usage: bash tools/profile_round.sh <tag>        e.g. r2_1gb
set -u
TAG=${1:-r2_1gb}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT $R/profiles
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extra --no-big --no-touched-pass --encode-sentences 2000000"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
cd $R
python tools/pmc_summary.py kernel-stats $OUT/stats profiles/${TAG}_kernel_stats.csv
python tools/pmc_summary.py pmc $OUT/fetch $OUT/write profiles/${TAG}_pmc_hbm.json
python bench.py ${BENCH_ARGS:-} > profiles/${TAG}_bench.json 2> $OUT/bench.err   # (BENCH_ARGS: e.g. "--cpu-runs 1" when GPU minutes are short)
cp profiles/${TAG}_*.* $R/gpurun_out/ 2>/dev/null
tail -c 600 profiles/${TAG}_bench.json
