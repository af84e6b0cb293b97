#!/bin/bash
# GPU tuning aid: rocprofv3 kernel + copy trace of ONE training; the launches before the merge loop (front end), with the gaps between them.
# usage: bash tools/dbg/frontend_trace.sh TAG KIND MB
TAG=$1; KIND=$2; MB=$3
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ft_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr -- python $R/tools/dbg/short_train.py $KIND $MB 32000 > $OUT/log.txt 2>&1
python - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("yttm::", "")) for r in csv.DictReader(open(f))]
for g in glob.glob(out + "/tr/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "(copy " + r.get("Direction", "?").replace("MEMORY_COPY_", "") + ")"))
rows.sort()
# the training's front end: from the first k_scan_bytes to the first merge-apply kernel
i0 = next(i for i, r in enumerate(rows) if r[2].startswith("k_scan_bytes"))
i1 = next(i for i, r in enumerate(rows) if i > i0 and (r[2].startswith("k_tiles<512") and ", true, " in r[2]))
t0 = rows[i0][0]
prev = t0
with open(out + "/frontend.txt", "w") as o:
    for s, e, k in rows[i0:i1 + 1]:
        o.write("%9.1f us  +gap %7.1f  dur %8.1f  %s\n" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, k[:70]))
        prev = e
print(open(out + "/frontend.txt").read())
PY
rm -rf $OUT/tr
