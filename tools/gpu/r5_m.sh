#!/bin/bash
# round 5: the chunked front end with the next chunk's upload under the kernels (landing buffer + device-to-device copy) against upload-then-work;
# the 100 MB pins in chunks; the beyond-2^32 block of the bench (8 GB Zipf, 8.8 GB heavy word: whole / serial chunks / overlapped chunks)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -x -q -k "chunked or smoke or train_matches" 2>&1 | tail -5 ) > gpurun_out/r5_m_test.log
cat gpurun_out/r5_m_test.log | tail -3
( timeout 1500 python bench.py --steps 3 --warmup 1 --no-encode --no-cpu-baseline --no-extra --big-zipf --no-touched-pass > gpurun_out/r5_m_big_bench.json ) 2> gpurun_out/r5_m_big.err
tail -3 gpurun_out/r5_m_big.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/r5_m_big_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
for k, v in d["extra"].items():
    print(k, v.get("seconds"), v.get("all_seconds"), v.get("chunked_front_end"))
print({k: v for k, v in d["parity"].items() if v is not True})
P
