#!/bin/bash
# round 4, GPU call F: tile record regions A/B (K4 by ranges of rounds), Zipf unaffected?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/dbg/ab_k4.py gpurun_out/r4f_ab.json abcd+zipf 1000 -- recs: norecs:YTTM_TILE_RECS=0 > gpurun_out/r4f_ab.log 2>&1
python - <<'P'
import json
for l in open("gpurun_out/r4f_ab.log"):
    if l.startswith(("abcd ", "zipf ")):
        kind, name = l.split()[:2]; d = json.loads(l.split(" ", 2)[2])
        print(kind, name, d["wall_s"], d["kernels_ms"]["merge_apply"], d["kernels_ms"].get("cand_scan"), d["k4_ms_by_rounds[sum,avg_us]"], d["matches_pin"])
P
tail -3 gpurun_out/r4f_ab.log | cut -c1-300
