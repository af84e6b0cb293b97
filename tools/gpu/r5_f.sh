#!/bin/bash
# round 5, call 6: GPU tests with the three-byte front end; CJK / abcd again; class-A grid cap on Zipf
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r5f_gputest.log
grep -n "passed\|failed\|rror" gpurun_out/r5f_gputest.log | head -5
( timeout 600 python tools/dbg/ab_k4.py gpurun_out/r5f_ab_cjk.json cjk 1000 -- base: ) > gpurun_out/r5f_ab_cjk.log 2>&1
( timeout 600 python tools/dbg/ab_k4.py gpurun_out/r5f_ab_abcd.json abcd 1000 -- base: ) > gpurun_out/r5f_ab_abcd.log 2>&1
( timeout 600 python tools/dbg/ab_k4.py gpurun_out/r5f_ab_zipf.json zipf 1000 -- base: g128:YTTM_APPLY_GRID=128 g384:YTTM_APPLY_GRID=384 g512:YTTM_APPLY_GRID=512 ) > gpurun_out/r5f_ab_zipf.log 2>&1
python - <<P
import json
for c in ("cjk","abcd","zipf"):
    d=json.load(open("gpurun_out/r5f_ab_%s.json" % c))
    for k,v in d.items(): print(k, v["wall_s"], v["rounds"], v["seconds_merge"], v["seconds_frontend"], v["kernels_ms"], v["matches_pin"])
P
