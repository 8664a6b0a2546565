"""Per-kernel instruction counts of a bench step from the two SQ counter passes of scripts/pmc_sq.sh:
python scripts/sq_summary.py gpurun_out/pmc_sq/runc/N_counter_collection.csv gpurun_out/pmc_sq2/runc/M_counter_collection.csv
Only the FIRST dispatch of each hot kernel after the warm-up is taken for the split kernels (one launch of level 2 and of
the count kernel is a whole step; level 1's launches of the step are summed)."""
import csv, sys, collections
HOT = ("kc_l1_reads16_kernel", "kc_l1_reads_kernel", "kc_l2_rec6_kernel", "kc_l2_split_kernel", "kc_count_kernel")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = next((h for h in HOT if h in r["Kernel_Name"]), None)
        if k is None:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[(k, f)].add(r["Dispatch_Id"])
wave_records = 6.5e9 / 64
for k in HOT:
    v = agg[k]
    if not any(kk == k for (kk, f) in ndisp):
        continue
    nd = max(len(s) for (kk, f), s in ndisp.items() if kk == k)
    steps = nd / (5 if "l1_reads" in k else 1)   # the stage step and the end-to-end legs run the same kernels
    print("%s: %d dispatches = %.0f passes over 50 M reads" % (k, nd, steps))
    per = lambda c: v.get(c, 0) / steps / wave_records
    print("   per 64 k-mers: VALU %.1f  SALU %.1f  LDS %.1f  VMEM read %.2f  VMEM write %.2f  SMEM %.2f" %
          (per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_LDS"), per("SQ_INSTS_VMEM_RD"), per("SQ_INSTS_VMEM_WR"), per("SQ_INSTS_SMEM")))
    if v.get("SQ_LDS_IDX_ACTIVE"):
        print("   LDS: %.3g active cycles, %.0f %% of them bank conflicts" % (v["SQ_LDS_IDX_ACTIVE"] / steps, 100 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]))
    if v.get("SQ_WAVE_CYCLES"):
        print("   of the waves' cycles: %.0f %% issuing, %.0f %% waiting for an instruction's operands (s_waitcnt), %.0f %% waiting in all" %
              (100 * v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
