#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "pinned_chunks" > gpurun_out/p_tests.log 2>&1
tail -3 gpurun_out/p_tests.log
timeout 900 python tools/dbg/encode_h2h.py > gpurun_out/p_h2h.log 2>&1
grep -E "best|Error|error|host -> host" gpurun_out/p_h2h.log | tail -40
cat /sys/kernel/mm/transparent_hugepage/enabled; nproc
