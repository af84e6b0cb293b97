"""GPU tuning aid: HBM traffic of the merge loop's kernels by ranges of merge rounds, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
ONE training (tools/dbg/short_train.py).  usage: python tools/dbg/pmc_by_rounds.py <fetch_dir> <write_dir>
A round is counted like tools/dbg/round_trace.sh counts it: one class-A apply launch (k_tiles<512, .., true, ..> or k_words<..>) per round, from
the pair count on.  Read bytes are given raw (counter x 1024) -- what the x2 correction of a wide streaming read would make of them is the
calibration's business (tools/micro/pmc_calib.hip), not this table's."""
import collections, csv, glob, sys


def dispatches(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("yttm::", ""), float(r["Counter_Value"]) * 1024.0,
                         int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    return rows


bounds = [(1, 11), (12, 21), (22, 46), (47, 100), (101, 200), (201, 300), (301, 450), (451, 10 ** 9)]
fetch, write = dispatches(sys.argv[1], "FETCH_SIZE"), dispatches(sys.argv[2], "WRITE_SIZE")
assert [k for _, k, _, _ in fetch] == [k for _, k, _, _ in write], "the two passes launched different kernels"
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0, 0]))
rnd, started = 0, False
for (_, k, rd, ns), (_, _, wr, _) in zip(fetch, write):
    if k.startswith("k_pair_count"):
        started, rnd = True, 1
        continue
    if not started:
        continue
    b = next(x for x in bounds if x[0] <= rnd <= x[1])
    e = per[b][k]
    e[0] += 1; e[1] += rd; e[2] += wr; e[3] += ns
    if (k.startswith("k_tiles<512") and ", true, " in k) or k.startswith("k_words<"):
        rnd += 1
for b in bounds:
    if b not in per:
        continue
    n_rounds = min(b[1], rnd - 1) - b[0] + 1
    tot_r = sum(e[1] for e in per[b].values()); tot_w = sum(e[2] for e in per[b].values())
    print("rounds %d-%d (%d): read %.2f MB raw + written %.2f MB per round, all kernels" % (b[0], min(b[1], rnd - 1), n_rounds, tot_r / n_rounds / 1e6, tot_w / n_rounds / 1e6))
    for k, e in sorted(per[b].items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:5]:
        print("   %-44s launches %5d  read raw %9.3f MB  written %9.3f MB per launch  (%.1f us under the counters)" % (k[:44], e[0], e[1] / e[0] / 1e6, e[2] / e[0] / 1e6, e[3] / e[0] / 1e3))
