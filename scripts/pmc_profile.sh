# rocprofv3 PMC passes of the default bench workload (one step, no warm-up, no CPU / end-to-end legs), counters in
# separate passes as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains
# beside --kernel-trace).  Writes gpurun_out/pmc_<pass>/ and the per-kernel summary gpurun_out/pmc_summary.json
# (copy it to profiles/rNN_pmc_50Mreads.json).  Usage: bash scripts/pmc_profile.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample-reads 0 --no-end-to-end"
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/pmc_$1 -- $B > $R/gpurun_out/pmc_$1.log 2>&1 && echo "$1 ok" || { echo "$1 FAILED"; exit 1; }; }
rm -rf $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_sq2 $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_l2 $R/gpurun_out/pmc_grbm
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" || exit 1
run sq2 "SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" || exit 1
run fetch "FETCH_SIZE" || exit 1
run write "WRITE_SIZE" || exit 1
run l2 "TCC_HIT_sum TCC_MISS_sum" || exit 1
run grbm "GRBM_GUI_ACTIVE" || exit 1
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(dict)
for d in ("sq", "sq2", "fetch", "write", "l2", "grbm"):
    for f in glob.glob("$R/gpurun_out/pmc_%s/*/*_counter_collection.csv" % d):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            agg[k].update(v)
    # kernel durations of the same pass (for the record; the bench's own HIP-event times are the ones quoted)
for f in glob.glob("$R/gpurun_out/pmc_fetch/*/*_kernel_trace.csv"):
    dur = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        dur[k][0] += 1
        dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k, (n, ms) in dur.items():
        if k in agg:
            agg[k]["launches"] = n
            agg[k]["total_ms_under_pmc"] = ms
# the clock the chip held: GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
for f in glob.glob("$R/gpurun_out/pmc_grbm/*/*_kernel_trace.csv"):
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0].replace("void ", "")] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9
    for k, sec in dur.items():
        if k in agg and "GRBM_GUI_ACTIVE" in agg[k] and sec > 0:
            agg[k]["effective_clock_GHz"] = agg[k]["GRBM_GUI_ACTIVE"] / 8.0 / sec / 1e9
json.dump(agg, open("$R/gpurun_out/pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, v in agg.items():
    if any(x in k for x in ("l1_reads", "l2_split", "l2_rec6", "count_kernel")):
        print(k[-48:], {a: "%.3g" % b for a, b in sorted(v.items())})
PY
