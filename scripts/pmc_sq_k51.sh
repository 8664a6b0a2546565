R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sq51
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_sq51 -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample-reads 0 --no-end-to-end --k 51 --reads 30000000 > $R/gpurun_out/pmc_sq51.log 2>&1 && echo ok
