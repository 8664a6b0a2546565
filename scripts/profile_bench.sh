# rocprofv3 kernel-trace summary of the default bench command; writes gpurun_out/prof_bench/
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py "$@" > $R/gpurun_out/prof_bench.log 2>&1
echo exit=$?
tail -1 $R/gpurun_out/prof_bench.log | cut -c1-600
