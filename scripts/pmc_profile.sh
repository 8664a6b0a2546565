R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample-reads 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_sq -- $B > $R/gpurun_out/pmc_sq.log 2>&1 && echo sq ok
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_sq2 -- $B > $R/gpurun_out/pmc_sq2.log 2>&1 && echo sq2 ok
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- $B > $R/gpurun_out/pmc_fetch.log 2>&1 && echo fetch ok
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- $B > $R/gpurun_out/pmc_write.log 2>&1 && echo write ok
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_l2 -- $B > $R/gpurun_out/pmc_l2.log 2>&1 && echo l2 ok
ls $R/gpurun_out/pmc_sq/*/
