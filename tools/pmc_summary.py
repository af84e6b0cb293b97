#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into the small CSV/JSON files kept under profiles/.

    python tools/pmc_summary.py kernel-stats <dir> <out.csv>         # --kernel-trace --stats run
    python tools/pmc_summary.py pmc <fetch_dir> <write_dir> <out.json>  # two --pmc passes (FETCH_SIZE, WRITE_SIZE)

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1 KB per count (TCC_EA0 requests x 64 B / 1024);
on gfx950 FETCH_SIZE under-reports wide (16 B/lane) coalesced streaming reads by exactly 2x
(/opt/skills/guides/MI355X_MICROARCH.md, section HBM), so both the raw value and the x2-corrected read side are given."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def source_sha16(root=None):
    """Hash of the product's sources (youtokentome_amd/csrc, include/): what a profile under profiles/ is a profile OF.  The GPU box has
    no .git, so this -- not a commit id -- is what pmc() records and what bench.py compares before it quotes a file's traffic."""
    import hashlib
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    files = []
    for d in ("youtokentome_amd/csrc", "include"):
        for f in sorted(os.listdir(os.path.join(root, d))):
            if f.endswith((".hip", ".h", ".cpp", ".c")) or f == "Makefile":
                files.append(os.path.join(d, f))
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    n = name.split("(")[0]
    return n.replace("void ", "").replace("yttm::", "")


def kernel_stats(d, out):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(out, "w") as o:
        o.write("kernel,calls,total_ms,avg_us,pct,min_us,max_us\n")  # (names are quoted: template arguments hold commas)
        for r in rows:
            o.write("\"%s\",%s,%.3f,%.3f,%s,%.3f,%.3f\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                      float(r["AverageNs"]) / 1e3, r["Percentage"], float(r["MinNs"]) / 1e3,
                                                      float(r["MaxNs"]) / 1e3))


def pmc(fetch_dir, write_dir, out):
    res = defaultdict(lambda: {"launches": 0, "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0, "ns": 0})
    for d, counter in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            res[k][counter + "_KB"] += float(r["Counter_Value"])
            if counter == "FETCH_SIZE":
                res[k]["launches"] += 1
                res[k]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    outd = {}
    for k, v in res.items():
        n = max(1, v["launches"])
        rd, wr = v["FETCH_SIZE_KB"] * 1024, v["WRITE_SIZE_KB"] * 1024
        outd[k] = {"launches": v["launches"], "avg_us": round(v["ns"] / n / 1e3, 2),
                   "hbm_read_bytes_per_launch_raw": round(rd / n), "hbm_read_bytes_per_launch_x2": round(2 * rd / n),
                   "hbm_write_bytes_per_launch": round(wr / n),
                   "traffic_bytes_per_launch": round((2 * rd + wr) / n)}
    outd["_meta"] = {"source_sha16": source_sha16(), "what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; source_sha16 = tools/pmc_summary.py source_sha16() of the build profiled"}
    json.dump(outd, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "kernel-stats":
        kernel_stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
