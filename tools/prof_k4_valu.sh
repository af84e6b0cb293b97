#!/bin/bash
# Per-launch instruction counts of the K4 apply kernel next to the per-round tile statistics (tuning aid).
# usage (GPU box): bash tools/prof_k4_valu.sh   -> gpurun_out/valu_rounds.txt, gpurun_out/valu_pmc/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-encode"
YTTM_TRACE_ROUNDS=$R/gpurun_out/valu_rounds.txt timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
  --kernel-include-regex "k_tiles" --output-format csv -d $R/gpurun_out/valu_pmc -- $CMD > $R/gpurun_out/valu_pmc.log 2>&1
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
rows = collections.defaultdict(dict)
for f in glob.glob(R + "/gpurun_out/valu_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "Lb1E" in r["Kernel_Name"] or "true" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
with open(R + "/gpurun_out/valu_per_launch.txt", "w") as o:
    for d in sorted(rows):
        o.write("%d %s\n" % (d, " ".join("%s=%.0f" % kv for kv in sorted(rows[d].items()))))
print(len(rows), "launches")
PY
