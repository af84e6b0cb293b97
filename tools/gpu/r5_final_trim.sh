#!/bin/bash
# round 5, after a change of the kernels late in the round: the evidence that names the build (GPU suite with the full-size pins, configs[1] with its
# PMC passes and the default bench line, the CJK-shaped corpus' kernel stats and PMC passes) -- (round 5's full script, r5_final.sh, is gone) without the blocks that do not
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out profiles
export TMPDIR=/tmp
( YTTM_FULL_PINS=1 timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > profiles/r5_gputest.log
cp profiles/r5_gputest.log gpurun_out/
grep -n "passed\|failed\|rror" profiles/r5_gputest.log | head -3
if ! grep -q " passed" profiles/r5_gputest.log || grep -q "failed" profiles/r5_gputest.log; then echo "GPU SUITE NOT GREEN: stopping"; exit 1; fi
BENCH_ARGS="--cpu-runs 1" timeout 900 bash tools/profile_round.sh r5_1gb abcd > gpurun_out/r5_profile_1gb.log 2>&1
tail -2 gpurun_out/r5_profile_1gb.log | cut -c1-300
timeout 240 bash tools/profile_round.sh r5_cjk cjk > gpurun_out/r5_profile_cjk.log 2>&1
head -4 profiles/r5_cjk_kernel_stats.csv
cp profiles/r5_1gb_* profiles/r5_cjk_* gpurun_out/ 2>/dev/null
python - <<'P'
import json
d = json.loads(open("profiles/r5_1gb_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "bad parity", [k for k, v in d["parity"].items() if v is False])
for k, v in d["extra"].items():
    print(k, v.get("ms_per_step"), v.get("us_per_round"))
print("dropout", d["encode_dropout"]["value"], "encode", d["encode"]["value"])
P
