#!/bin/bash
# round 5, call 5: scan_top after the refine rewrite -- phases (PROF build) and plain A/B on Zipf / abcd
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in zipf abcd; do
  ( YTTM_AMD_LIB=$PWD/youtokentome_amd/libyttm_prof.so timeout 600 python tools/dbg/ab_k4.py gpurun_out/r5e_prof_$c.json $c 1000 -- base: ) > gpurun_out/r5e_prof_$c.log 2>&1
  grep "scan_top\|fused rounds" gpurun_out/r5e_prof_$c.log | tail -2
  ( timeout 600 python tools/dbg/ab_k4.py gpurun_out/r5e_ab_$c.json $c 1000 -- base: ) > gpurun_out/r5e_ab_$c.log 2>&1
  grep "fused rounds\|merge loop\|rounds 1501-\|rounds 451-" gpurun_out/r5e_ab_$c.log | tail -4
  python - <<P
import json
d=json.load(open("gpurun_out/r5e_ab_$c.json"))
for k,v in d.items(): print(k, v["wall_s"], v["rounds"], v["seconds_merge"], v["seconds_frontend"], v["kernels_ms"], v["matches_pin"])
P
done
