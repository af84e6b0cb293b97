#!/bin/bash
# the front end under the upload: part size and K2b grid sweep (file -> model wall, YTTM_TRACE lines)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "front_end_under" > gpurun_out/s_tests.log 2>&1
tail -2 gpurun_out/s_tests.log
timeout 900 python tools/dbg/fe_overlap.py > gpurun_out/s_sweep.log 2>&1
grep -E "^(cfg|\[yttm\] front)" gpurun_out/s_sweep.log | tail -60
