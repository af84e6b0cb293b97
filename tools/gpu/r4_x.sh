#!/bin/bash
# last evidence refresh of the round: smoke, the GPU suite, the profiled command's kernel stats, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out profiles
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/x_smoke.log 2>&1; tail -1 gpurun_out/x_smoke.log
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 ) > gpurun_out/r4_gputest.log
cat gpurun_out/r4_gputest.log
timeout 2400 bash tools/profile_round.sh r4_1gb > gpurun_out/r4_profile_round.log 2>&1
cp profiles/r4_1gb_* gpurun_out/
python - <<'P'
import json
d = json.loads(open("profiles/r4_1gb_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "hbm", d.get("value_hbm_resident"), "encode", d["encode"]["value"], d["encode"]["ms_per_step"], d["encode"].get("value_host_to_host"), "dropout", d["encode_dropout"]["value"])
print("parity", all(v is not False for v in d["parity"].values()), [k for k, v in d["parity"].items() if v is False])
P
