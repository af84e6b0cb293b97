#!/bin/bash
# K5: one word per lane + length classes, A/B on the GPU; then the kernel times of the default build
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "one_word_per_lane or k5_word_cache" > gpurun_out/l_tests.log 2>&1
tail -3 gpurun_out/l_tests.log
timeout 900 python tools/dbg/encode_ab.py > gpurun_out/l_ab.log 2>&1
cat gpurun_out/l_ab.log | tail -14
