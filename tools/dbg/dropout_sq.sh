cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; mkdir -p gpurun_out/r5sq; O=$R/gpurun_out/r5sq
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- python $R/tools/dbg/dropout_ab.py 2000000 -- base: > $O/log.txt 2>&1
grep "kernel ms" $O/log.txt
python - $O <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/sq/**/*counter_collection.csv", recursive=True)[0]
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "k5_encode" in r["Kernel_Name"]:
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for d in sorted(rows)[-2:]:
    print(d, {k: "%.4g" % v for k, v in sorted(rows[d].items())})
P
rm -rf $O/sq
