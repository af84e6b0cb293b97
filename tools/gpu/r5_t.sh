#!/bin/bash
# round 5, last call: the GPU test-suite with the full-size pins at the head (profiles/r5_gputest.log), then random corpora on the MI355X for the
# minutes that are left: word mode against the tiles (tools/soak_gpu_words.py), the encoder under random hooks incl. BPE-dropout's differential
# check (tools/soak_encode.py with the product library)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( YTTM_FULL_PINS=1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r5_gputest.log
grep -n "passed\|failed\|rror" gpurun_out/r5_gputest.log | head -3
timeout 330 python tools/soak_gpu_words.py 240 41 2>&1 | tail -2 | tee gpurun_out/r5_soak_gpu_words.log
YTTM_AMD_LIB=$PWD/youtokentome_amd/libyttm_mi355x.so timeout 330 python tools/soak_encode.py 240 42 2>&1 | tail -2 | tee gpurun_out/r5_soak_gpu_encode.log
