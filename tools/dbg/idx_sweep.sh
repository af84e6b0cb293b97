#!/bin/bash
# GPU tuning aid: K4 / candidate-family times for a few settings of the pair-index hooks
for cfg in "64 4" "128 4" "128 6" "64 8" "256 3" "32 8"; do
  set -- $cfg
  echo "post_per_tile=$1 sparse_div=$2: $(YTTM_INDEX_POST_PER_TILE=$1 YTTM_INDEX_SPARSE_DIV=$2 python tools/dbg/frontend_time.py abcd 1000 2>/dev/null | tail -1)"
done
echo "no index: $(YTTM_NO_INDEX=1 python tools/dbg/frontend_time.py abcd 1000 2>/dev/null | tail -1)"
