#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "k5 or encode or long_words" > gpurun_out/z_tests.log 2>&1
tail -2 gpurun_out/z_tests.log
VARIANTS="3" KIND=abcd bash tools/gpu/r4_m.sh 2>&1 | grep -E "^abcd|k5"
cd $R; timeout 600 python tools/dbg/encode_ab.py 10000000 zipf 3 2>&1 | grep "^zipf"
