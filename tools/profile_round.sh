#!/bin/bash
# Regenerates the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun from the repo root):
#   kernel-trace --stats of one bench step, FETCH_SIZE and WRITE_SIZE in separate --pmc passes (counters never share a run with trace
#   domains other than the kernel trace), and -- for the default workload -- the bench line itself.
# usage: bash tools/profile_round.sh <tag> [abcd|zipf|cjk|encode]        e.g. r5_1gb abcd ; r5_zipf zipf ; r5_cjk cjk ; r5_encode10m encode
#   abcd   BASELINE configs[1] (+ the encode of 2e6 sentences), then the full default bench line -> profiles/<tag>_bench.json
#   zipf   configs[2] as the main workload, train only            cjk   the CJK-shaped corpus, train only
#   encode configs[3]/[4]: the 10 M-sentence encode (with and without dropout) behind one training
# The PMC summary records the hash of the sources it was taken of (tools/pmc_summary.py source_sha16); bench.py quotes it only for that build.
set -u
TAG=${1:-r5_1gb}
WHAT=${2:-abcd}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT $R/profiles
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extra --no-big --no-touched-pass"
case $WHAT in
  abcd)   CMD="python $R/bench.py $COMMON --encode-sentences 2000000" ;;
  zipf)   CMD="python $R/bench.py $COMMON --corpus zipf --no-encode" ;;
  cjk)    CMD="python $R/bench.py $COMMON --corpus cjk --no-encode" ;;
  encode) CMD="python $R/bench.py $COMMON --encode-sentences 10000000" ;;
  *) echo "unknown workload $WHAT"; exit 2 ;;
esac
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
cd $R
python tools/pmc_summary.py kernel-stats $OUT/stats profiles/${TAG}_kernel_stats.csv
python tools/pmc_summary.py pmc $OUT/fetch $OUT/write profiles/${TAG}_pmc_hbm.json
if [ "$WHAT" = abcd ]; then
  T0=$(date +%s)
  python bench.py ${BENCH_ARGS:-} > profiles/${TAG}_bench.json 2> $OUT/bench.err   # (BENCH_ARGS: e.g. "--cpu-runs 1" when GPU minutes are short)
  echo "default bench.py run: $(( $(date +%s) - T0 )) s of wall clock"
  tail -c 600 profiles/${TAG}_bench.json
fi
cp profiles/${TAG}_*.* $R/gpurun_out/ 2>/dev/null
head -12 profiles/${TAG}_kernel_stats.csv
