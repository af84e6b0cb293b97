#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 ) > gpurun_out/r4_gputest.log
cat gpurun_out/r4_gputest.log
timeout 600 python tools/dbg/encode_h2h.py > gpurun_out/w_h2h.log 2>&1
grep -E "best|sub-batches" gpurun_out/w_h2h.log | tail -12
