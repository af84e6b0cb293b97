#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
for v in 2 3; do VARIANTS="$v" KIND=abcd bash $R/tools/gpu/r4_m.sh 2>&1 | grep -E "^abcd|k5_words|k5w_list|k5w_count_slots|k5w_insert"; done
cd $R; timeout 600 python tools/dbg/encode_ab.py 10000000 zipf 2,3 2>&1 | grep "^zipf lanes"
