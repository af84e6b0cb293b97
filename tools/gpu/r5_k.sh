#!/bin/bash
# round 5, call 11: the word-mode switch rule on abcd / CJK / zipf4m (YTTM_WORD_DIV), GPU suite timing with ten ordering runs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python tools/dbg/ab_k4.py gpurun_out/r5k_ab_abcd.json abcd 1000 -- base: d100:YTTM_WORD_DIV=100 d120:YTTM_WORD_DIV=120 d150:YTTM_WORD_DIV=150 ) > gpurun_out/r5k_ab_abcd.log 2>&1
( timeout 900 python tools/dbg/ab_k4.py gpurun_out/r5k_ab_cjk.json cjk 1000 -- base: d120:YTTM_WORD_DIV=120 d300:YTTM_WORD_DIV=300 ) > gpurun_out/r5k_ab_cjk.log 2>&1
python - <<P
import json
for c in ("abcd","cjk"):
    d=json.load(open("gpurun_out/r5k_ab_%s.json" % c))
    for k,v in d.items(): print(k, v["wall_s"], v["rounds"], v["seconds_merge"], v["word_switch_round"], v["kernels_ms"]["merge_apply"], v["kernels_ms"]["cand_scan"], v["matches_pin"])
P
( time timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) 2>&1 | tail -6
