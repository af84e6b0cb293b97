#!/bin/bash
# round 5: BPE-dropout -- the GPU tests that touch it, then configs[4] (10^7 sentences of 128 chars, p = 0.1) by sentences per pack
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q -k "dropout or encode" 2>&1 | tail -5 ) | tee gpurun_out/r5_o_test.log
timeout 900 python tools/dbg/dropout_ab.py 10000000 -- base: s2:YTTM_DROPOUT_PACK_SENT=2 s1:YTTM_DROPOUT_PACK_SENT=1 g1:YTTM_K5_GROUP=1 2>&1 | grep -v "^\[yttm\]\|^id: " | tee gpurun_out/r5_o_ab.log
