#!/bin/bash
# round 5, call 13: the GPU test-suite with the full-size pins (profiles/r5_gputest.log)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( YTTM_FULL_PINS=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r5_gputest.log
grep -n "passed\|failed\|rror" gpurun_out/r5_gputest.log | head -5
