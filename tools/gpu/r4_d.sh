#!/bin/bash
# round 4, GPU call D: the direct pair table A/B; the corpora past 2^32
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( YTTM_K4_DIRECT=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-encode --no-cpu-baseline --no-extra --no-big --no-touched-pass > gpurun_out/r4d_bench_nodirect.json ) 2> gpurun_out/r4d_bench_nodirect.err
( timeout 1500 python bench.py --steps 5 --warmup 2 --no-encode --no-cpu-baseline --no-extra --no-touched-pass > gpurun_out/r4d_bench.json ) 2> gpurun_out/r4d_bench.err
python - <<'P'
import json
for n in ("r4d_bench_nodirect.json", "r4d_bench.json"):
    try:
        d = json.loads(open("gpurun_out/" + n).read().strip().splitlines()[-1])
        print(n, "value", d["value"], "ms", d["ms_per_step"], "hbm", d.get("value_hbm_resident"), d["hbm_resident"]["ms_per_step"], "parity", d["parity"])
        print("  kernels", {k: (v["ms_total"], v["launches"]) for k, v in d["kernels"].items()})
        for k, v in d.get("extra", {}).items():
            print("  extra", k, json.dumps(v)[:900])
    except Exception as e:
        print(n, "unreadable:", e)
P
tail -5 gpurun_out/r4d_bench.err
