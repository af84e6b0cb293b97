#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python bench.py --steps 3 --warmup 1 --no-encode --no-cpu-baseline --no-big --no-touched-pass --no-extra2 > gpurun_out/r4h_bench.json ) 2> gpurun_out/r4h_bench.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/r4h_bench.json").read().strip().splitlines()[-1])
print("abcd value", d["value"], "ms", d["ms_per_step"], "hbm", d.get("value_hbm_resident"), d["hbm_resident"]["ms_per_step"])
print("  kernels", {k: (v["ms_total"], v["launches"], v["GBps"]) for k, v in d["kernels"].items()}, d["phases_s"])
for k, v in d.get("extra", {}).items():
    print(k, "value", v["value"], "ms", v["ms_per_step"], "us/round", v.get("us_per_round"), v["phases_s"])
    print("  kernels", {kk: (vv["ms_total"], vv["launches"]) for kk, vv in v["kernels"].items()})
print(d["parity"])
P
