# rocprofv3 kernel-trace summary of the default bench command INCLUDING its end-to-end legs (host-resident input: the
# level-2 instalments show as many short kc_l2_split_kernel launches); writes gpurun_out/prof_bench_e2e/ -- copy the
# *_kernel_stats.csv to profiles/rNN_bench_default_end_to_end_kernel_stats.csv
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench_e2e
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_e2e -- python3 $R/bench.py --cpu-sample-reads 0 "$@" > $R/gpurun_out/prof_bench_e2e.log 2>&1
echo exit=$?
tail -1 $R/gpurun_out/prof_bench_e2e.log | cut -c1-300
