#!/bin/bash
# round 5, call 7: BPE-dropout with packed links / live test by rule against round 4's scheme (same seed: same ids); CJK front end again
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python tools/dbg/dropout_ab.py 10000000 -- new: old:YTTM_DROPOUT_NO_PACK=1 ) > gpurun_out/r5g_dropout.log 2>&1
tail -3 gpurun_out/r5g_dropout.log
( timeout 600 python tools/dbg/ab_k4.py gpurun_out/r5g_ab_cjk.json cjk 1000 -- base: ) > gpurun_out/r5g_ab_cjk.log 2>&1
python - <<P
import json
d=json.load(open("gpurun_out/r5g_ab_cjk.json"))
for k,v in d.items(): print(k, v["wall_s"], v["rounds"], v["seconds_merge"], v["seconds_frontend"], v["kernels_ms"], v["matches_pin"])
P
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "dropout or golden_train or long_words or very_long" 2>&1 | tail -3 )
