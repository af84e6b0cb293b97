#!/bin/bash
# round 4: the evidence kept under profiles/ -- GPU test-suite, rocprofv3 kernel stats + PMC passes + the full bench line, the forced-communicator
# bench and its timeline, the 8 GB block
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out profiles
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r4_gputest.log
tail -3 gpurun_out/r4_gputest.log
timeout 2400 bash tools/profile_round.sh r4_1gb > gpurun_out/r4_profile_round.log 2>&1
( YTTM_BENCH_FORCE_COMM=1 timeout 900 python bench.py --steps 10 --warmup 3 --no-encode --no-cpu-baseline --no-big --no-extra2 > profiles/r4_forced_comm_bench.json ) 2> gpurun_out/r4_fc.err
timeout 600 bash tools/dbg/round_trace.sh fc_final abcd 1000 comm > gpurun_out/r4_fc_trace.log 2>&1
cp gpurun_out/rt_fc_final/summary.txt profiles/r4_forced_comm_rounds.txt
timeout 600 bash tools/dbg/round_trace.sh plain_final abcd 1000 > gpurun_out/r4_plain_trace.log 2>&1
cp gpurun_out/rt_plain_final/summary.txt profiles/r4_1gb_trace_by_rounds.txt
rm -rf gpurun_out/rt_fc_final/tr gpurun_out/rt_plain_final/tr
( timeout 1200 python bench.py --steps 3 --warmup 1 --no-encode --no-cpu-baseline --no-extra --big-zipf --no-touched-pass > profiles/r4_big_bench.json ) 2> gpurun_out/r4_big.err
cp profiles/r4_* gpurun_out/
python - <<'P'
import json
for n in ("r4_1gb_bench.json", "r4_forced_comm_bench.json", "r4_big_bench.json"):
    try:
        d = json.loads(open("profiles/" + n).read().strip().splitlines()[-1])
        print(n, "value", d["value"], "ms", d["ms_per_step"], "hbm", d.get("value_hbm_resident"), "parity", all(v is not False for v in d["parity"].values()), [k for k, v in d["parity"].items() if v is False])
    except Exception as e:
        print(n, "unreadable:", e)
P
