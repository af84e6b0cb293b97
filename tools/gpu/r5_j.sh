#!/bin/bash
# round 5, call 10: the whole default bench line once (reduced CPU baseline) -- does every new block run?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python bench.py --steps 3 --warmup 1 --cpu-runs 1 > gpurun_out/r5j_bench.json ) 2> gpurun_out/r5j_bench.err
echo "rc=$?"
python - <<'P'
import json
d = json.loads(open("gpurun_out/r5j_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "hbm", d.get("value_hbm_resident"), "roofline", d["roofline"]["frac"], d["roofline"].get("traffic_source", "")[:80])
print("config", {k: d["config"].get(k) for k in ("merge_rounds", "batch_splits", "batch_extensions", "top_refills", "repacks", "front_end_under_the_upload")})
print("e2e", json.dumps(d.get("e2e", {}))[:1200])
print("encode", d["encode"]["value"], d["encode"]["kernel_ms"], "dropout", d["encode_dropout"]["value"], d["encode_dropout"]["kernel_ms"])
for k, v in d.get("extra", {}).items(): print(k, v.get("ms_per_step") or v.get("seconds"), v.get("us_per_round"), json.dumps(v.get("chunked_front_end"))[:300] if v.get("chunked_front_end") else "")
print("parity", d["parity"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("train_seconds"))
P
tail -3 gpurun_out/r5j_bench.err
