#!/bin/bash
# round 5, VERDICT r4 item 1 (c): what the merge loop's HBM traffic is made of.  (1) the counters calibrated on known byte counts for the kernels'
# access patterns (tools/micro/pmc_calib.hip); (2) FETCH_SIZE / WRITE_SIZE of ONE 1 GB abcd training by ranges of merge rounds; (3) the same
# training's per-round merge sites, words visited and tokens streamed (YTTM_TRACE) to set the bytes against.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out/r5n
O=$R/gpurun_out/r5n
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- $R/tools/micro/pmc_calib > $O/cal.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cal_write -- $R/tools/micro/pmc_calib >> $O/cal.log 2>&1
python - $O <<'P' | tee $O/calibration.txt
import csv, glob, sys
o = sys.argv[1]
known = {"calib_stream16": (4 << 30, "B streamed"), "calib_stream4": (4 << 30, "B streamed"), "calib_stream1": (1 << 30, "B streamed"), "calib_gather<8>": (1 << 24, "accesses"),
         "calib_gather<16>": (1 << 24, "accesses"), "calib_gather<64>": (1 << 24, "accesses"), "calib_gather<128>": (1 << 24, "accesses"), "calib_atomic8": (1 << 24, "accesses"), "calib_scatter8": (1 << 24, "accesses"),
         "calib_store16": (4 << 30, "B stored"), "calib_fill": (4 << 30, "B stored")}
res = {}
for d, c in (("cal_fetch", "FETCH_SIZE"), ("cal_write", "WRITE_SIZE")):
    f = glob.glob(o + "/" + d + "/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            res.setdefault(k, {})[c] = float(r["Counter_Value"]) * 1024.0
            res[k]["us_" + c] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, unit) in known.items():
    v = res.get(k, {})
    print("%-18s %12d %-10s FETCH_SIZE %14.0f B = %8.3f per unit   WRITE_SIZE %14.0f B = %8.3f per unit   %9.1f us" % (k, n, unit, v.get("FETCH_SIZE", -1), v.get("FETCH_SIZE", -1) / n,
          v.get("WRITE_SIZE", -1), v.get("WRITE_SIZE", -1) / n, v.get("us_FETCH_SIZE", -1)))
P
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/tools/dbg/short_train.py abcd 1000 32000 > $O/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/tools/dbg/short_train.py abcd 1000 32000 > $O/write.log 2>&1
cd $R
python tools/dbg/pmc_by_rounds.py $O/fetch $O/write | tee $O/traffic_by_rounds.txt
YTTM_TRACE=1 timeout 300 python tools/dbg/short_train.py abcd 1000 32000 2>&1 | grep "rounds \|merge loop" | tee $O/trace.txt
rm -rf $O/fetch $O/write $O/cal_fetch $O/cal_write
