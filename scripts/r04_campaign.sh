# The round's measurement campaign on one box (writes gpurun_out/r4/final/; copy what is to be judged into profiles/r04_*):
# the default bench line, its rocprofv3 kernel stats, the PMC passes, the longer k of BASELINE configs 4 and 5 with their
# kernel stats, SQ counters for k=51, the N > 1 code path at N = 1.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4/final
mkdir -p $O
cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo "bench default done"; tail -c 200 $O/bench_default.json
bash scripts/profile_bench.sh > $O/profile_bench.txt 2>&1
cp gpurun_out/prof_bench/*/*_kernel_stats.csv $O/bench_default_kernel_stats.csv 2>/dev/null
grep -h "^{\"metric\"" gpurun_out/prof_bench.log > $O/bench_default_under_rocprof.json
echo "kernel stats done"
bash scripts/pmc_profile.sh > $O/pmc_profile.txt 2>&1
cp gpurun_out/pmc_summary.json $O/pmc_50Mreads.json
python3 scripts/sq_summary.py gpurun_out/pmc_sq/*/*_counter_collection.csv gpurun_out/pmc_sq2/*/*_counter_collection.csv > $O/sq_insts.txt 2>&1
echo "pmc done"
cd /tmp && export TMPDIR=/tmp
for k in 33 51 55 77; do
  rm -rf $R/gpurun_out/prof_k$k
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_k$k -- python3 $R/bench.py --k $k --reads 30000000 --steps 3 --warmup 1 --cpu-sample-reads 0 --no-end-to-end > $O/bench_k$k.log 2>&1
  grep -h "^{\"metric\"" $O/bench_k$k.log > $O/bench_k$k.json
  cp $R/gpurun_out/prof_k$k/*/*_kernel_stats.csv $O/bench_k${k}_kernel_stats.csv 2>/dev/null
  echo "k=$k done"
done
rm -rf $R/gpurun_out/pmc_k51a $R/gpurun_out/pmc_k51b $R/gpurun_out/pmc_k51c $R/gpurun_out/pmc_k51d
B51="python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample-reads 0 --no-end-to-end --k 51 --reads 30000000"
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_k51a -- $B51 > $O/pmc_k51a.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_k51b -- $B51 > $O/pmc_k51b.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_k51c -- $B51 > $O/pmc_k51c.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc_k51d -- $B51 > $O/pmc_k51d.log 2>&1
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(dict)
for d in "abcd":
    for f in glob.glob("$R/gpurun_out/pmc_k51%s/*/*_counter_collection.csv" % d):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            agg[k].update(v)
json.dump({k: v for k, v in agg.items() if any(x in k for x in ("l1_reads", "l2_split", "l2_rec6", "count_kernel"))}, open("$O/pmc_k51_30Mreads.json", "w"), indent=1, sort_keys=True)
PY
echo "pmc k51 done"
cd $R
python3 bench.py --force-sharded --steps 3 --warmup 1 --cpu-sample-reads 0 --no-end-to-end > $O/bench_force_sharded.json 2> $O/bench_force_sharded.err
python3 bench.py --force-sharded --shard-flow records --steps 3 --warmup 1 --cpu-sample-reads 0 --no-end-to-end > $O/bench_force_sharded_records.json 2> $O/bench_force_sharded_records.err
echo "sharded done"
ls $O
