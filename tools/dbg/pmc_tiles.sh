#!/bin/bash
# GPU tuning aid: instruction / cycle counters of the first launches of the tile kernels (K3 and the dense K4 rounds) of one 1 GB training.
# usage (GPU box): bash tools/dbg/pmc_tiles.sh  -> gpurun_out/pmc_tiles.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python $R/tools/dbg/frontend_time.py abcd 1000"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
  --kernel-include-regex "${KREGEX:-k_tiles}" --output-format csv -d $R/gpurun_out/pmc_tiles_a -- $CMD > $R/gpurun_out/pmc_tiles_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAVES \
  --kernel-include-regex "${KREGEX:-k_tiles}" --output-format csv -d $R/gpurun_out/pmc_tiles_b -- $CMD > $R/gpurun_out/pmc_tiles_b.log 2>&1
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
rows = collections.defaultdict(dict)
names = {}
for tag in "ab":
    for f in glob.glob(R + "/gpurun_out/pmc_tiles_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            d = int(r["Dispatch_Id"])
            rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            names[d] = r["Kernel_Name"][:60]
with open(R + "/gpurun_out/pmc_tiles.txt", "w") as o:
    for d in sorted(rows)[:40]:
        o.write("%d %s %s\n" % (d, names[d], " ".join("%s=%.0f" % kv for kv in sorted(rows[d].items()))))
print(len(rows), "launches")
PY
rm -rf $R/gpurun_out/pmc_tiles_a $R/gpurun_out/pmc_tiles_b
