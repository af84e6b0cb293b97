#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "one_word_per_lane or k5_word_cache or golden_encode or mixed_shapes or long_words" > gpurun_out/n_tests.log 2>&1
tail -3 gpurun_out/n_tests.log
timeout 900 python tools/dbg/encode_ab.py 10000000 abcd,zipf ${VARIANTS:-2,3} > gpurun_out/n_ab.log 2>&1
grep -E "^(abcd|zipf) " gpurun_out/n_ab.log
