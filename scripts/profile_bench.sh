# rocprofv3 kernel-trace summary of the default bench command (same workload as the BENCH line); writes
# gpurun_out/prof_bench/ -- copy the *_kernel_stats.csv to profiles/rNN_bench_default_kernel_stats.csv
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --cpu-sample-reads 0 --no-end-to-end "$@" > $R/gpurun_out/prof_bench.log 2>&1
echo exit=$?
tail -1 $R/gpurun_out/prof_bench.log | cut -c1-400
