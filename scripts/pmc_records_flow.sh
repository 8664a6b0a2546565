# rocprofv3 PMC passes of the records flow in wire units at N = 1 (bench.py --force-sharded --shard-flow records, 50 M reads
# in one block, one step): the SQ counters and the HBM bytes of kc_bin16_kernel and kc_l1_wire6_kernel (csrc/kc_wire6.hpp),
# counters in separate passes, no trace domains beside --kernel-trace.  Writes gpurun_out/pmc_rf_<pass>/ and
# gpurun_out/pmc_records_flow.json / .txt (copy to profiles/).  Usage: bash scripts/pmc_records_flow.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --force-sharded --shard-flow records --block-reads 50000000 --steps 1 --warmup 0 --cpu-sample-reads 0 --no-end-to-end"
run() { rm -rf $R/gpurun_out/pmc_rf_$1; rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/pmc_rf_$1 -- $B > $R/gpurun_out/pmc_rf_$1.log 2>&1 && echo "$1 ok" || { echo "$1 FAILED"; exit 1; }; }
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" || exit 1
run sq2 "SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" || exit 1
run fetch "FETCH_SIZE" || exit 1
run write "WRITE_SIZE" || exit 1
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(dict)
for d in ("sq", "sq2", "fetch", "write"):
    for f in glob.glob("$R/gpurun_out/pmc_rf_%s/*/*_counter_collection.csv" % d):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            agg[k].update(v)
for f in glob.glob("$R/gpurun_out/pmc_rf_fetch/*/*_kernel_trace.csv"):
    dur = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        dur[k][0] += 1
        dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k, (n, ms) in dur.items():
        if k in agg:
            agg[k]["launches"] = n
            agg[k]["total_ms_under_pmc"] = ms
hot = {k: v for k, v in agg.items() if any(x in k for x in ("bin16", "l1_wire6", "l2_rec6", "count_kernel"))}
json.dump(hot, open("$R/gpurun_out/pmc_records_flow.json", "w"), indent=1, sort_keys=True)
w = 6.4e9 / 64  # wave-records of the step (occurrences with two neighbours)
with open("$R/gpurun_out/pmc_records_flow.txt", "w") as o:
    for k, v in sorted(hot.items()):
        per = lambda c: v.get(c, 0) / w
        o.write("%s: %d launches, %.1f ms under the counters\n" % (k, v.get("launches", 0), v.get("total_ms_under_pmc", 0)))
        o.write("   per 64 records: VALU %.1f  SALU %.1f  LDS %.1f  VMEM read %.2f  VMEM write %.2f\n" % (per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_LDS"), per("SQ_INSTS_VMEM_RD"), per("SQ_INSTS_VMEM_WR")))
        if v.get("SQ_LDS_IDX_ACTIVE"):
            o.write("   LDS: %.3g active cycles, %.0f %% of them bank conflicts\n" % (v["SQ_LDS_IDX_ACTIVE"], 100 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]))
        if v.get("SQ_WAVE_CYCLES"):
            o.write("   of the waves' cycles: %.0f %% issuing, %.0f %% waiting for operands (s_waitcnt), %.0f %% waiting in all\n" % (100 * v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            o.write("   HBM: read %.1f GB (FETCH_SIZE doubled, MI355X_MICROARCH.md), write %.1f GB\n" % (2 * v["FETCH_SIZE"] * 1024 / 1e9, v["WRITE_SIZE"] * 1024 / 1e9))
print(open("$R/gpurun_out/pmc_records_flow.txt").read())
PY
