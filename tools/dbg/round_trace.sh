#!/bin/bash
# GPU tuning aid: rocprofv3 kernel + memory-copy trace of ONE training, per-kernel durations and the gaps between them by ranges of merge rounds.
# usage: bash tools/dbg/round_trace.sh TAG KIND MB [comm] [ENV=VAL ...]
TAG=$1; KIND=$2; MB=$3; shift 3
COMM=""
if [ "$1" == "comm" ]; then COMM=comm; shift; fi
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/rt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr -- python $R/tools/dbg/short_train.py $KIND $MB 32000 $COMM > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("yttm::", "")) for r in csv.DictReader(open(f))]
for g in glob.glob(out + "/tr/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "(copy " + r.get("Direction", "?").replace("MEMORY_COPY_", "") + ")"))
rows.sort()
rnd = 0
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
bounds = [(1, 11), (12, 28), (29, 46), (47, 100), (101, 200), (201, 300), (301, 450), (451, 10 ** 9)]
def bucket(r):
    for a, b in bounds:
        if a <= r <= b: return (a, b)
started = False
prev_end = None
wall = collections.defaultdict(float)
for s, e, k in rows:
    if k.startswith("k_pair_count"): started = True; rnd = 1; prev_end = e; continue
    if not started: continue
    b = bucket(rnd)
    per[b][k][0] += 1; per[b][k][1] += (e - s) / 1e3
    g = "(gap before " + k.split("<")[0] + ")"
    per[b][g][0] += 1; per[b][g][1] += max(0, s - prev_end) / 1e3
    wall[b] += (e - prev_end) / 1e3
    prev_end = max(prev_end, e)
    if (k.startswith("k_tiles<512") and ", true, " in k) or k.startswith("k_words<"): rnd += 1  # (one per round: the class-A apply launch)
with open(out + "/summary.txt", "w") as o:
    for b in bounds:
        if b not in per: continue
        nr = min(b[1], rnd - 1) - b[0] + 1
        o.write("rounds %d-%d (%d): %.1f us per round (kernels + gaps)\n" % (b[0], min(b[1], rnd - 1), nr, wall[b] / max(nr, 1)))
        for k, (n, us) in sorted(per[b].items(), key=lambda kv: -kv[1][1]):
            o.write("   %-44s calls %6d  total %9.1f us  per round %8.2f us  avg %8.2f us\n" % (k[:44], n, us, us / max(nr, 1), us / n))
print(open(out + "/summary.txt").read())
PY
