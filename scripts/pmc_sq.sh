R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample-reads 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc_sq -- $B > $R/gpurun_out/pmc_sq.log 2>&1 && echo sq ok
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_sq2 -- $B > $R/gpurun_out/pmc_sq2.log 2>&1 && echo sq2 ok
python3 - <<PY
import csv,glob,collections
for d in ['pmc_sq','pmc_sq2']:
    f=sorted(glob.glob('$R/gpurun_out/'+d+'/*/*_counter_collection.csv'))[-1]
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-30:]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items():
        if any(x in k for x in ['l1_reads','l2_split','l2_rec6','count_kernel']): print(k, {a:'%.3g'%b for a,b in sorted(v.items())})
PY
