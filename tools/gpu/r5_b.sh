#!/bin/bash
# round 5, call 2: the GPU test-suite (with the reference's binding / stress harness through the bpe.h adapter), then the three pinned
# corpora again: scan_top's first slots with the list's length, the first tile with its header, class-B repack looks bounded
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r5b_gputest.log
tail -4 gpurun_out/r5b_gputest.log
for c in abcd zipf cjk; do
  ( timeout 600 python tools/dbg/ab_k4.py gpurun_out/r5b_ab_$c.json $c 1000 -- base: ) > gpurun_out/r5b_ab_$c.log 2>&1
  grep "rounds \|merge loop\|fused rounds" gpurun_out/r5b_ab_$c.log | tail -14
  python - <<P
import json
d=json.load(open("gpurun_out/r5b_ab_$c.json"))
for k,v in d.items(): print(k, v["wall_s"], v["rounds"], v["seconds_merge"], v["kernels_ms"], v["matches_pin"])
P
done
