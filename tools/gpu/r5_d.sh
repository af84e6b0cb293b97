#!/bin/bash
# round 5, call 4: GPU test-suite on the refactored build; scan_top with wave-aggregated LDS atomics (PROF marks + plain A/B);
# forced-communicator bench with the zipf_sharded block
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r5d_gputest.log
grep -n "passed\|failed\|rror" gpurun_out/r5d_gputest.log | head -5
( YTTM_AMD_LIB=$PWD/youtokentome_amd/libyttm_prof.so timeout 600 python tools/dbg/ab_k4.py gpurun_out/r5d_prof_zipf.json zipf 1000 -- base: ) > gpurun_out/r5d_prof_zipf.log 2>&1
grep "scan_top\|fused rounds" gpurun_out/r5d_prof_zipf.log | tail -2
for c in zipf abcd cjk; do
  ( timeout 600 python tools/dbg/ab_k4.py gpurun_out/r5d_ab_$c.json $c 1000 -- base: ) > gpurun_out/r5d_ab_$c.log 2>&1
  grep "fused rounds\|merge loop\|rounds 1501-\|rounds 451-" gpurun_out/r5d_ab_$c.log | tail -4
  python - <<P
import json
d=json.load(open("gpurun_out/r5d_ab_$c.json"))
for k,v in d.items(): print(k, v["wall_s"], v["rounds"], v["seconds_merge"], v["seconds_frontend"], v["kernels_ms"], v["matches_pin"])
P
done
( YTTM_BENCH_FORCE_COMM=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-encode --no-cpu-baseline --no-big --no-extra2 --no-touched-pass > gpurun_out/r5d_fc_bench.json ) 2> gpurun_out/r5d_fc.err
python - <<P
import json
d = json.loads(open("gpurun_out/r5d_fc_bench.json").read().strip().splitlines()[-1])
print("fc value", d["value"], "ms", d["ms_per_step"], {k: d["config"].get(k) for k in ("rccl_ranks","exchange_repeats","exchange_us_per_round","multi_gpu_mode","batch_splits")})
for k in ("zipf","zipf_sharded"):
    z = d["extra"][k]; print(k, z["ms_per_step"], z["us_per_round"], z["config"].get("multi_gpu_mode"), z["config"].get("exchange_us_per_round"), z["config"].get("exchange_repeats"))
print("parity", d["parity"])
P
tail -2 gpurun_out/r5d_fc.err
