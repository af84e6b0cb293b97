#!/bin/bash
# round 5: BPE-dropout (profiles/r5_dropout.txt), where the time goes: by sentences per pack, and by probability -- p = 1 (every word ends at its first pop: tokenization, word starts, the pairs' rules, the words'
# first events, compaction, output), p = 0.5, p = 0.1, p = 1e-18 (no event is ever skipped: every merge of the plain encoder through the queues).
# (Measured on the way, with builds that are gone: the nested loop -- a lane finishes its word before any takes the next -- 108.7 ms against the
# flat loop's 105.5 at three sentences per pack; a pop of the sorted array's front by head++ instead of a shift 105.5 -> 95.4; events compared
# packed + two-multiply draws 95.4 -> 91.9; heaps from 8 / 16 / 24 tokens on: 125 / 126 / 109 ms.)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python tools/dbg/dropout_ab.py 10000000 -- base: s2:YTTM_DROPOUT_PACK_SENT=2 s1:YTTM_DROPOUT_PACK_SENT=1 g1:YTTM_K5_GROUP=1 p1:P=1.0 p05:P=0.5 p01:P=0.1 p0:P=1e-18 p1s1:P=1.0,YTTM_DROPOUT_PACK_SENT=1 2>&1 | grep -v "^\[yttm\]\|^id: " | tee gpurun_out/r5_p_ab.log
