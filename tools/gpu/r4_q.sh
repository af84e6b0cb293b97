#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "dropout" > gpurun_out/q_tests.log 2>&1
tail -3 gpurun_out/q_tests.log
timeout 900 python tools/dbg/encode_ab.py 10000000 abcd,zipf 2 > gpurun_out/q_ab.log 2>&1
grep -E "^(abcd|zipf) " gpurun_out/q_ab.log
