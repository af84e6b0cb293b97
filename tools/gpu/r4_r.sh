#!/bin/bash
# the front end under the upload: GPU test, then the file -> model step with and without it
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "front_end_under or golden_train or medium_corpus" > gpurun_out/r_tests.log 2>&1
tail -3 gpurun_out/r_tests.log
for v in 0 1; do
  if [ $v = 1 ]; then export YTTM_FE_NO_OVERLAP=1; else unset YTTM_FE_NO_OVERLAP; fi
  timeout 900 python bench.py --steps 4 --warmup 1 --no-encode --no-cpu-baseline --no-big --no-extra --no-touched-pass > gpurun_out/r_bench_$v.json 2> gpurun_out/r_bench_$v.err
  python - <<P
import json
d = json.loads(open("gpurun_out/r_bench_$v.json").read().strip().splitlines()[-1])
print("no_overlap=$v value", d["value"], "ms", d["ms_per_step"], "hbm", d.get("value_hbm_resident"), d["phases_s"], "parity", d["parity"])
print({k: (v["ms_total"], v["launches"]) for k, v in d["kernels"].items()})
P
done
