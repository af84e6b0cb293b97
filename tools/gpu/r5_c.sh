#!/bin/bash
# round 5, call 3: the candidate scan's phases (PROF build marks), the forced-communicator step with the front end under the upload,
# CJK with the class-B looks spaced out
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in zipf abcd; do
  ( YTTM_AMD_LIB=$PWD/youtokentome_amd/libyttm_prof.so timeout 600 python tools/dbg/ab_k4.py gpurun_out/r5c_prof_$c.json $c 1000 -- base: ) > gpurun_out/r5c_prof_$c.log 2>&1
  grep "scan_top\|fused rounds\|merge loop" gpurun_out/r5c_prof_$c.log | tail -6
done
( timeout 600 python tools/dbg/ab_k4.py gpurun_out/r5c_ab_cjk.json cjk 1000 -- base: ) > gpurun_out/r5c_ab_cjk.log 2>&1
grep "merge loop\|rounds 1501\|rounds 701" gpurun_out/r5c_ab_cjk.log | tail -4
python - <<P
import json
d=json.load(open("gpurun_out/r5c_ab_cjk.json"))
for k,v in d.items(): print(k, v["wall_s"], v["rounds"], v["seconds_merge"], v["seconds_frontend"], v["kernels_ms"], v["matches_pin"])
P
( YTTM_BENCH_FORCE_COMM=1 timeout 900 python bench.py --steps 5 --warmup 2 --no-encode --no-cpu-baseline --no-big --no-extra2 --no-touched-pass > gpurun_out/r5c_fc_bench.json ) 2> gpurun_out/r5c_fc.err
( timeout 900 python bench.py --steps 5 --warmup 2 --no-encode --no-cpu-baseline --no-big --no-extra2 --no-touched-pass > gpurun_out/r5c_plain_bench.json ) 2> gpurun_out/r5c_plain.err
python - <<P
import json
for n in ("r5c_fc_bench.json", "r5c_plain_bench.json"):
    try:
        d = json.loads(open("gpurun_out/" + n).read().strip().splitlines()[-1])
        print(n, "value", d["value"], "ms", d["ms_per_step"], "hbm ms", d.get("hbm_resident", {}).get("ms_per_step"), "fe overlapped", d["config"].get("front_end_under_the_upload"), "zipf ms", d.get("extra", {}).get("zipf", {}).get("ms_per_step"), d.get("extra", {}).get("zipf", {}).get("config", {}).get("multi_gpu_mode"), "parity", [k for k, v in d["parity"].items() if v is False], d.get("e2e", {}).get("train_first_call"))
    except Exception as e:
        print(n, "unreadable:", e)
P
tail -3 gpurun_out/r5c_fc.err
