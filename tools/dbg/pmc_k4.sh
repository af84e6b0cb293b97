#!/bin/bash
# GPU tuning aid: per-launch instruction / cycle counters of the K4 apply kernels over the first rounds of a 1 GB training.
# usage (GPU box): bash tools/dbg/pmc_k4.sh TAG [VOCAB] [KIND]   (env hooks such as YTTM_K4_OLD=1 pass through)  -> gpurun_out/pmc_k4_TAG.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; VOCAB=${2:-60}; KIND=${3:-abcd}
CMD="python $R/tools/dbg/short_train.py $KIND 1000 $VOCAB"
KRE="k_apply_pm|k_tilesILi512"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
  --kernel-include-regex "$KRE" --output-format csv -d $R/gpurun_out/pmc_k4_a -- $CMD > $R/gpurun_out/pmc_k4_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAIT_ANY \
  --kernel-include-regex "$KRE" --output-format csv -d $R/gpurun_out/pmc_k4_b -- $CMD > $R/gpurun_out/pmc_k4_b.log 2>&1
TAG=$TAG python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
rows = collections.defaultdict(dict)
names, dur = {}, {}
for tag in "ab":
    for f in glob.glob(R + "/gpurun_out/pmc_k4_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            d = int(r["Dispatch_Id"])
            rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            names[d] = r["Kernel_Name"][:40]
            dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
with open(R + "/gpurun_out/pmc_k4_%s.txt" % os.environ["TAG"], "w") as o:
    for d in sorted(rows)[:60]:
        o.write("%d %s us=%.1f %s\n" % (d, names[d], dur[d], " ".join("%s=%.0f" % kv for kv in sorted(rows[d].items()))))
print(len(rows), "launches")
PY
rm -rf $R/gpurun_out/pmc_k4_a $R/gpurun_out/pmc_k4_b
