cd /tmp && export TMPDIR=/tmp
R=/root/repo
CMD="python $R/bench.py --size-mb 200 --steps 1 --warmup 0 --no-cpu-baseline --encode-sentences 4000000"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-include-regex k5_encode --output-format csv -d $R/gpurun_out/pmc_enc_$i -- $CMD > $R/gpurun_out/pmc_enc_$i.log 2>&1
done
python - <<'PY'
import csv,glob,collections
for i in range(1,5):
    for f in glob.glob('/root/repo/gpurun_out/pmc_enc_%d/**/*counter_collection.csv'%i, recursive=True):
        acc=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'][:30],r['Counter_Name'])]+=float(r['Counter_Value'])
        for k,v in sorted(acc.items()): print(k,v)
PY
