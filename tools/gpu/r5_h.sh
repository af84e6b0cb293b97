#!/bin/bash
# round 5, call 8: BPE-dropout -- unsorted bag against the sorted array (same seed: same ids)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python tools/dbg/dropout_ab.py 10000000 -- new: r4:YTTM_DROPOUT_NO_PACK=1 ) > gpurun_out/r5h_dropout.log 2>&1
tail -4 gpurun_out/r5h_dropout.log
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "dropout" 2>&1 | tail -3 )
