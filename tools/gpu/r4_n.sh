#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python tools/dbg/encode_ab.py 10000000 abcd,zipf 0,3,4 > gpurun_out/n_ab.log 2>&1
grep -E "^(abcd|zipf) " gpurun_out/n_ab.log
