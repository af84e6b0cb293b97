#!/bin/bash
# round 5: the evidence kept under profiles/ -- GPU test-suite incl. the full-size pins, rocprofv3 kernel stats + PMC passes of four workloads
# (configs[1] with the full bench line, configs[2], the CJK-shaped corpus, the 10 M-sentence encode), the forced-communicator bench, the timeline
# by ranges of rounds, the beyond-2^32 block.  (tools/gpu/r5_[a-k].sh are the round's measurement calls in order; their outputs are quoted in
# DESIGN.md, the ones kept are under profiles/.)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out profiles
export TMPDIR=/tmp
( YTTM_FULL_PINS=1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > profiles/r5_gputest.log
grep -n "passed\|failed\|rror" profiles/r5_gputest.log | head -3
timeout 1500 bash tools/profile_round.sh r5_1gb abcd > gpurun_out/r5_profile_1gb.log 2>&1
tail -2 gpurun_out/r5_profile_1gb.log | cut -c1-300
for w in zipf cjk; do
  timeout 600 bash tools/profile_round.sh r5_$w $w > gpurun_out/r5_profile_$w.log 2>&1
  head -4 profiles/r5_${w}_kernel_stats.csv
done
timeout 900 bash tools/profile_round.sh r5_encode10m encode > gpurun_out/r5_profile_encode.log 2>&1
head -6 profiles/r5_encode10m_kernel_stats.csv
( YTTM_BENCH_FORCE_COMM=1 timeout 900 python bench.py --steps 5 --warmup 2 --no-encode --no-cpu-baseline --no-big --no-extra2 --no-touched-pass > profiles/r5_forced_comm_bench.json ) 2> gpurun_out/r5_fc.err
timeout 600 bash tools/dbg/round_trace.sh plain_final abcd 1000 > gpurun_out/r5_plain_trace.log 2>&1
cp gpurun_out/rt_plain_final/summary.txt profiles/r5_1gb_trace_by_rounds.txt
rm -rf gpurun_out/rt_plain_final/tr
( timeout 1500 python bench.py --steps 3 --warmup 1 --no-encode --no-cpu-baseline --no-extra --big-zipf --no-touched-pass > profiles/r5_big_bench.json ) 2> gpurun_out/r5_big.err
cp profiles/r5_* gpurun_out/
python - <<'P'
import json
for n in ("r5_1gb_bench.json", "r5_forced_comm_bench.json", "r5_big_bench.json"):
    try:
        d = json.loads(open("profiles/" + n).read().strip().splitlines()[-1])
        print(n, "value", d["value"], "ms", d["ms_per_step"], "hbm", d.get("value_hbm_resident"), "frac", (d.get("roofline") or {}).get("frac"), "traffic", (d.get("roofline") or {}).get("traffic"), "parity ok", all(v is not False for v in d["parity"].values()), [k for k, v in d["parity"].items() if v is False])
    except Exception as e:
        print(n, "unreadable:", e)
P
