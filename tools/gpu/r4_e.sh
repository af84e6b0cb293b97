#!/bin/bash
# round 4, GPU call E: K4 by ranges of rounds (direct table on/off), per-phase cycle shares of the tile kernel (PROF=2 build)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/dbg/ab_k4.py gpurun_out/r4e_ab.json abcd 1000 -- direct: nodirect:YTTM_K4_DIRECT=0 > gpurun_out/r4e_ab.log 2>&1
timeout 600 python tools/dbg/k4_phases.py 1000 > gpurun_out/r4e_phases.txt 2>&1
YTTM_K4_DIRECT=0 timeout 600 python tools/dbg/k4_phases.py 1000 > gpurun_out/r4e_phases_nodirect.txt 2>&1
python - <<'P'
import json
for l in open("gpurun_out/r4e_ab.log"):
    if l.startswith("abcd "):
        name = l.split()[1]; d = json.loads(l.split(" ", 2)[2])
        print(name, d["wall_s"], d["kernels_ms"]["merge_apply"], d["k4_ms_by_rounds[sum,avg_us]"], d["matches_pin"])
P
grep "^rounds" gpurun_out/r4e_phases.txt | head -8
echo ---- no direct
grep "^rounds" gpurun_out/r4e_phases_nodirect.txt | head -4
