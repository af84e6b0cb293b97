#!/bin/bash
# SQ counters of the K4 kernels (tuning aid; separate --pmc passes, no trace domains). usage: bash tools/prof_k4_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-encode"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "k_filter|k_tiles" --output-format csv -d $R/gpurun_out/pmc_k4_$i -- $CMD > $R/gpurun_out/pmc_k4_$i.log 2>&1
done
python - <<'PY'
import csv,glob,collections,os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for i in range(1,4):
    for f in glob.glob(R+'/gpurun_out/pmc_k4_%d/**/*counter_collection.csv'%i, recursive=True):
        acc=collections.defaultdict(float); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('yttm::','')
            acc[(k,r['Counter_Name'])]+=float(r['Counter_Value'])
        for k,v in sorted(acc.items()): print("%-32s %-24s %.4g"%(k[0],k[1],v))
PY
