#!/bin/bash
# round 4, GPU call C: ordering tests; forced-comm bench (main workload only); host-side split of a forced-comm training; its timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_tail_ordering or rccl_world" 2>&1 | tail -30 ) > gpurun_out/r4c_gputest.log
tail -3 gpurun_out/r4c_gputest.log
( YTTM_BENCH_FORCE_COMM=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-encode --no-cpu-baseline --no-extra > gpurun_out/r4c_bench_fc.json ) 2> gpurun_out/r4c_bench_fc.err
( timeout 600 python bench.py --steps 5 --warmup 2 --no-encode --no-cpu-baseline --no-extra --no-touched-pass > gpurun_out/r4c_bench.json ) 2> gpurun_out/r4c_bench.err
timeout 300 python tools/dbg/comm_train.py abcd 1000 > gpurun_out/r4c_comm_train.txt 2>&1
timeout 600 bash tools/dbg/round_trace.sh fc2 abcd 1000 comm > gpurun_out/r4c_trace_fc.txt 2>&1
rm -rf gpurun_out/rt_fc2/tr
python - <<'P'
import json
for n in ("r4c_bench.json", "r4c_bench_fc.json"):
    try:
        d = json.loads(open("gpurun_out/" + n).read().strip().splitlines()[-1])
        print(n, "value", d["value"], "ms", d["ms_per_step"], "hbm", d.get("value_hbm_resident"), d["hbm_resident"]["ms_per_step"], "parity", d["parity"])
        print("  kernels", {k: (v["ms_total"], v["launches"]) for k, v in d["kernels"].items()})
        print("  phases", d["phases_s"])
    except Exception as e:
        print(n, "unreadable:", e)
P
grep -a "merge loop wall\|wall " gpurun_out/r4c_comm_train.txt | tail -4
