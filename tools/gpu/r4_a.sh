#!/bin/bash
# round 4, GPU call A: the GPU test-suite, the bench line (file -> model), the same through a communicator of one rank
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/r4a_gputest.log
( timeout 900 python bench.py --steps 5 --warmup 2 --no-extra2 --cpu-runs 1 > gpurun_out/r4a_bench.json ) 2> gpurun_out/r4a_bench.err
( YTTM_BENCH_FORCE_COMM=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-encode --no-cpu-baseline --no-extra2 > gpurun_out/r4a_bench_fc.json ) 2> gpurun_out/r4a_bench_fc.err
tail -5 gpurun_out/r4a_gputest.log
python - <<'P'
import json
for n in ("r4a_bench.json", "r4a_bench_fc.json"):
    try:
        d = json.loads(open("gpurun_out/" + n).read().strip().splitlines()[-1])
        print(n, "value", d["value"], "ms", d["ms_per_step"], "hbm", d.get("value_hbm_resident"), "parity", d["parity"])
        print("  kernels", {k: (v["ms_total"], v["launches"]) for k, v in d["kernels"].items()})
        print("  phases", d["phases_s"], "zipf", d.get("extra", {}).get("zipf", {}).get("ms_per_step"))
    except Exception as e:
        print(n, "unreadable:", e)
P
