#!/bin/bash
# GPU tuning aid: rocprofv3 kernel trace of ONE training (1 GB abcd), per-kernel durations by ranges of merge rounds.
# usage: bash tools/dbg/words_trace.sh TAG [ENV=VAL ...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/wt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $R/tools/dbg/short_train.py abcd 1000 32000 > $OUT/log.txt 2>&1
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("yttm::", "")) for r in csv.DictReader(open(f))]
rows.sort()
rnd = 0
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
bounds = [(1, 11), (12, 28), (29, 46), (47, 100), (101, 200), (201, 300), (301, 450), (451, 10 ** 9)]
def bucket(r):
    for a, b in bounds:
        if a <= r <= b: return (a, b)
started = False
prev_end = None
for s, e, k in rows:
    if k.startswith("k_pair_count"): started = True; rnd = 1; prev_end = e; continue
    if not started: continue
    b = bucket(rnd)
    per[b][k][0] += 1; per[b][k][1] += (e - s) / 1e3
    per[b]["(gap before)"][0] += 1; per[b]["(gap before)"][1] += max(0, s - prev_end) / 1e3
    prev_end = e
    if k.startswith("k_tiles<512") or k.startswith("k_words<"): rnd += 1
with open(out + "/summary.txt", "w") as o:
    for b in bounds:
        if b not in per: continue
        nr = min(b[1], rnd - 1) - b[0] + 1
        o.write("rounds %d-%d (%d):\n" % (b[0], min(b[1], rnd - 1), nr))
        for k, (n, us) in sorted(per[b].items(), key=lambda kv: -kv[1][1]):
            o.write("   %-40s calls %6d  total %9.1f us  per round %8.2f us  avg %8.2f us\n" % (k[:40], n, us, us / max(nr, 1), us / n))
print(open(out + "/summary.txt").read())
PY
