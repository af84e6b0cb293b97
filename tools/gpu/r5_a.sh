#!/bin/bash
# round 5, call 1: where a round's wall time goes (host timeline by ranges of rounds, YTTM_TRACE) on the three pinned 1 GB corpora,
# the doorbell micro-benchmark (launch per round against a pre-launched gate / a resident kernel), a few list-size A/Bs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 120 tools/micro/doorbell 2000 ) > gpurun_out/r5a_doorbell.txt 2>&1
tail -30 gpurun_out/r5a_doorbell.txt
( timeout 600 python tools/dbg/ab_k4.py gpurun_out/r5a_ab_abcd.json abcd 1000 -- base: top512:YTTM_TOP_TARGET=512,YTTM_TOP_MIN=96 top256:YTTM_TOP_TARGET=256,YTTM_TOP_MIN=64 ) > gpurun_out/r5a_ab_abcd.log 2>&1
grep "rounds \|merge loop\|fused rounds\|host pick" gpurun_out/r5a_ab_abcd.log | tail -60
( timeout 600 python tools/dbg/ab_k4.py gpurun_out/r5a_ab_zipf.json zipf 1000 -- base: top512:YTTM_TOP_TARGET=512,YTTM_TOP_MIN=96 top256:YTTM_TOP_TARGET=256,YTTM_TOP_MIN=64 ) > gpurun_out/r5a_ab_zipf.log 2>&1
grep "rounds \|merge loop\|fused rounds\|host pick" gpurun_out/r5a_ab_zipf.log | tail -60
( timeout 600 python tools/dbg/ab_k4.py gpurun_out/r5a_ab_cjk.json cjk 1000 -- base: ) > gpurun_out/r5a_ab_cjk.log 2>&1
grep "rounds \|merge loop\|fused rounds\|host pick" gpurun_out/r5a_ab_cjk.log | tail -30
