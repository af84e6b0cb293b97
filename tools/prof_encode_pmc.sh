#!/bin/bash
# SQ counters of the batch encoder's kernels (word-cache path: k5w_insert, k5_words, k5w_count, k5w_scatter; direct path: k5_encode) on the
# bench's 1e7 sentences: instruction mix and what the waves wait for.  Counters in passes of their own (never with trace domains other than
# the kernel trace).  usage (GPU box): bash tools/prof_encode_pmc.sh  -> profiles/r4_encode_sq_counters.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/dbg/encode_ab.py 10000000 abcd 2 1"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmc_enc_$i
  timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "k5" --output-format csv -d /tmp/pmc_enc_$i -- $CMD > /tmp/pmc_enc_$i.log 2>&1
done
mkdir -p $R/gpurun_out
python - > $R/gpurun_out/r4_encode_sq_counters.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float)
calls = collections.Counter()
for i in range(1, 5):
    for f in glob.glob('/tmp/pmc_enc_%d/**/*counter_collection.csv' % i, recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:40]
            acc[(k, r['Counter_Name'])] += float(r['Counter_Value'])
            key = (k, r['Dispatch_Id'])
            if i == 1 and key not in seen:
                seen.add(key)
                calls[k] += 1
print("# rocprofv3 --pmc passes of: tools/dbg/encode_ab.py 10000000 abcd 2 1 (1e7 sentences of 128 random 'abcd ' chars; sums over the launches of the run:")
print("# the direct path and the word-cache path once warm-up + once timed each, then three dropout launches)")
ks = sorted({k for k, _ in acc})
for k in ks:
    print("%s  (%d launches)" % (k, calls[k]))
    for (kk, c), v in sorted(acc.items()):
        if kk == k:
            print("    %-26s %.4g" % (c, v))
PY
cat $R/gpurun_out/r4_encode_sq_counters.txt | head -120
