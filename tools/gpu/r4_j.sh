#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/dbg/ab_k4.py gpurun_out/r4j_ab.json abcd 1000 -- base: h64d150:YTTM_HOT_TARGET_WORDS=65536,YTTM_WORD_DIV=150 h64d120:YTTM_HOT_TARGET_WORDS=65536,YTTM_WORD_DIV=120 h64d100:YTTM_HOT_TARGET_WORDS=65536,YTTM_WORD_DIV=100 h64d80:YTTM_HOT_TARGET_WORDS=65536,YTTM_WORD_DIV=80 h96d120:YTTM_HOT_TARGET_WORDS=98304,YTTM_WORD_DIV=120 base2: > gpurun_out/r4j_ab.log 2>&1
python - <<'P'
import json
for l in open("gpurun_out/r4j_ab.log"):
    if l.startswith(("abcd ", "zipf ")):
        kind, name = l.split()[:2]; d = json.loads(l.split(" ", 2)[2])
        print(kind, name, "wall", d["wall_s"], "merge", d["seconds_merge"], d["kernels_ms"]["merge_apply"], d["kernels_ms"].get("cand_scan"), "idx", d["index_builds"], "switch", d["word_switch_round"], "rounds", d["rounds"], d["matches_pin"])
P
