#!/bin/bash
# GPU tuning aid: class-A apply grid for small tile sets (1 GB Zipf corpus: 6290 tiles)
for g in 768 512 384 256 192 128 96; do
  echo "grid=$g: $(YTTM_APPLY_GRID=$g python tools/dbg/trace_train.py zipf 1000 32000 2>&1 | grep -E 'train wall' | tail -1)"
done
