#!/bin/bash
# round 4, GPU call B: the ordering tests; kernel timeline of one training through a communicator of one rank and of a plain one
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_tail_ordering or rccl_world" 2>&1 | tail -30 ) > gpurun_out/r4b_gputest.log
tail -5 gpurun_out/r4b_gputest.log
timeout 600 bash tools/dbg/round_trace.sh fc abcd 1000 comm > gpurun_out/r4b_trace_fc.txt 2>&1
timeout 600 bash tools/dbg/round_trace.sh plain abcd 1000 > gpurun_out/r4b_trace_plain.txt 2>&1
rm -rf gpurun_out/rt_fc/tr gpurun_out/rt_plain/tr
