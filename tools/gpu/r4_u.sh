#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python bench.py --steps 4 --warmup 1 --no-encode --no-cpu-baseline --no-big --no-extra --no-touched-pass > gpurun_out/u_bench.json 2> gpurun_out/u_bench.err
python - <<P
import json
d = json.loads(open("gpurun_out/u_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "hbm", d.get("value_hbm_resident"), d["hbm_resident"]["ms_per_step"] if "hbm_resident" in d else None, d["phases_s"], "parity", d["parity"])
print({k: (v["ms_total"], v["launches"]) for k, v in d["kernels"].items()})
print(d["hbm_resident"].get("phases_s"), {k: (v["ms_total"], v["launches"]) for k, v in d["hbm_resident"].get("kernels", {}).items()})
P
