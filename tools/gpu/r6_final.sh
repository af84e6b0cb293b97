#!/bin/bash
# Round 6: the evidence that names the build (run through gpurun from the repo root; every file it leaves under profiles/ carries the hash of the
# sources it was taken of, and bench.py quotes a file only for that build).  usage: bash tools/gpu/r6_final.sh [suite|abcd|cjk|zipf|encode|comm|soak ...]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out profiles
export TMPDIR=/tmp
WHAT=${*:-suite abcd cjk zipf encode comm soak}
for w in $WHAT; do
  case $w in
    suite)
      ( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15 ) > profiles/r6_gputest.log
      cp profiles/r6_gputest.log gpurun_out/
      grep -n "passed\|failed\|rror" profiles/r6_gputest.log | head -3
      if ! grep -q " passed" profiles/r6_gputest.log || grep -q "failed" profiles/r6_gputest.log; then echo "GPU SUITE NOT GREEN: stopping"; exit 1; fi ;;
    abcd)
      BENCH_ARGS="${BENCH_ARGS:-}" timeout 1500 bash tools/profile_round.sh r6_1gb abcd > gpurun_out/r6_profile_1gb.log 2>&1
      tail -2 gpurun_out/r6_profile_1gb.log | cut -c1-300
      python - <<'P'
import json
d = json.loads(open("profiles/r6_1gb_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "bad parity", [k for k, v in d["parity"].items() if v is False])
for k, v in d["extra"].items():
    print(k, v.get("ms_per_step"), v.get("us_per_round"))
print("dropout", d["encode_dropout"]["value"], "encode", d["encode"]["value"])
P
      ;;
    cjk) timeout 400 bash tools/profile_round.sh r6_cjk cjk > gpurun_out/r6_profile_cjk.log 2>&1; head -6 profiles/r6_cjk_kernel_stats.csv ;;
    zipf) timeout 400 bash tools/profile_round.sh r6_zipf zipf > gpurun_out/r6_profile_zipf.log 2>&1; head -5 profiles/r6_zipf_kernel_stats.csv ;;
    encode) timeout 600 bash tools/profile_round.sh r6_encode10m encode > gpurun_out/r6_profile_encode.log 2>&1; grep "k5" profiles/r6_encode10m_kernel_stats.csv | head -8 ;;
    comm)
      YTTM_BENCH_FORCE_COMM=1 timeout 600 python bench.py --no-cpu-baseline --no-e2e --no-big > profiles/r6_forced_comm_bench.json 2> gpurun_out/r6_forced_comm.err
      python - <<'P'
import json
d = json.loads(open("profiles/r6_forced_comm_bench.json").read().strip().splitlines()[-1])
print("forced comm: value", d["value"], "ms", d["ms_per_step"], "hbm-resident ms", d.get("hbm_resident", {}).get("ms_per_step"), "rccl_ranks", d["config"].get("rccl_ranks"), "bad parity", [k for k, v in d["parity"].items() if v is False])
P
      ;;
    soak)
      timeout 300 python tools/soak_gpu_words.py 200 66 > gpurun_out/r6_soak_words.log 2>&1; tail -2 gpurun_out/r6_soak_words.log
      timeout 200 python tools/soak_gpu_words.py 120 67 cjk > gpurun_out/r6_soak_cjk.log 2>&1; tail -2 gpurun_out/r6_soak_cjk.log ;;
  esac
done
cp profiles/r6_* gpurun_out/ 2>/dev/null
