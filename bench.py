#!/usr/bin/env python3
"""bench.py -- k-mers/s of the kcount stage on MI355X (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = the whole kcount stage over one batch of synthetic ArcticSynth-shaped reads that is
already resident in HBM: table reset, extract + insert of every read (N>1: bin by owner shard,
exchange over RCCL, insert), vote/purge/compaction to dense result arrays in HBM.  At N=1 the
workload is BASELINE.json configs[1] (50 M reads of 150 bp, k=21); for N>1 every GPU gets the same
number of reads (weak scaling).  value = raw k-mers (sum of len-k+1, kcount.cpp:86) of all ranks
per second of the slowest rank.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_KMER = {1: 34.0, 2: 46.0, 3: 58.0, 4: 70.0}  # SURVEY.md section 8d contract constants, by num_longs
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
PROFILE = "profiles/r04_pmc_50Mreads.json"  # rocprofv3 PMC passes of this same command (scripts/pmc_profile.sh)


def own_alg_bytes(kernel, nl, k, read_len, results_per_raw):
    """Algorithmic HBM bytes per raw k-mer of ONE kernel of this design: what that kernel must
    read and write at least, so that its fraction of the roofline cannot exceed 1.  Compact records (k <= 23): level 1
    writes 8-byte records -- 6-byte ones where the remainder below the bucket fits 32 bits (kc_l1_reads16_kernel /
    kc_l2_rec6_kernel, k = 21) -- and level 2 turns them into 4-byte ones."""
    rec1 = 6.0 if ("l1_reads16" in kernel or "l2_rec6" in kernel) else 8.0 * nl
    rec2 = 4.0 if (nl == 1 and k <= 23) else 8.0 * nl
    if "l1_reads" in kernel:
        return 2.0 * read_len / (read_len - k + 1) + rec1   # bases + qualities in, level-1 records out
    if "l2_split" in kernel or "l2_rec6" in kernel:
        return rec1 + rec2
    if "count_kernel" in kernel:
        return rec2 + results_per_raw * (8.0 * nl + 4.0)    # region records in, dense results out
    return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=50_000_000, help="reads per GPU (150 bp each)")
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--block-reads", type=int, default=8_000_000, help="N>1: reads per exchange block")
    ap.add_argument("--cpu-sample-reads", type=int, default=4_000_000, help="reads timed through the CPU oracle (0 = skip)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-resident (PCIe-inclusive) leg at N=1")
    ap.add_argument("--check", action="store_true", help="size-independent result checks after the timed region")
    ap.add_argument("--table-path", action="store_true", help="A/B: force the global-table insert path instead of the bucketed one")
    ap.add_argument("--force-sharded", action="store_true", help="run the N>1 code path (extract per block, exchange, receive) even at N=1")
    ap.add_argument("--shard-flow", choices=["auto", "single-pass", "records"], default="auto",
                    help="N>1: single-pass = a shard owns level-1 buckets (kc_shard_*), records = hash ownership "
                         "(kc_extract_partition / kc_insert_records); auto = single-pass when the shard's share of the regions holds its k-mers")
    ap.add_argument("--tune", default="", help="A/B: bucketed-path geometry overrides, e.g. slots=2048,p1=1024,p2=1024")
    return ap.parse_args()


def pmc_traffic(nreads, k, tuned):
    """HBM bytes per STEP of every kernel from the committed rocprofv3 PMC passes of this same workload
    (scripts/pmc_profile.sh: FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for wide coalesced reads on gfx950).  {} when no profile matches the workload."""
    if nreads != 50_000_000 or k != 21 or tuned:
        return {}, None
    try:
        prof = json.load(open(os.path.join(ROOT, PROFILE)))
    except Exception:
        return {}, None
    out = {}
    for name, ctr in prof.items():
        if "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
            ent = {"bytes": (2.0 * ctr["FETCH_SIZE"] + ctr["WRITE_SIZE"]) * 1024.0}
            if ctr.get("TCC_HIT_sum", 0) + ctr.get("TCC_MISS_sum", 0) > 0:
                ent["l2_hit"] = ctr["TCC_HIT_sum"] / (ctr["TCC_HIT_sum"] + ctr["TCC_MISS_sum"])
            out[name] = ent
    return out, PROFILE


def usable_cores():
    """CPU threads this process may really use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(k, nreads, full_reads, params):
    """The oracle (a port of the reference CPU kcount, oracle/kcount_oracle.c: minimizer-hash partition into one
    emulated rank per core, ASCII supermers, receiver re-derives the k-mers, MurmurHash3 + prime-capacity linear-probe
    table with separate key and value arrays, two-scan finalize per rank, no sort) on this host's cores, on a bounded
    sample of the same read stream at the SAME depth as the GPU run: the genomes are scaled with the number of reads,
    and every rank's table is sized for the sample's distinct k-mers, so nothing rehashes while timed (`rehashes`).
    Reported beside the GPU number; never the target."""
    import mhm2_kmer_analysis_v2_amd as pkg
    from oracle import cpu_oracle as O
    cores = usable_cores()
    scale = nreads / float(full_reads)
    sp = pkg.synth_params(min_genome_len=max(1000, int(params.min_genome_len * scale)),
                          max_genome_len=max(2000, int(params.max_genome_len * scale)))
    b, q, offs = pkg.synth_reads_host(nreads, params=sp)
    L = sp.read_len
    raw = nreads * (L - k + 1)  # fixed-length reads
    distinct = sp.num_genomes * (sp.min_genome_len + sp.max_genome_len) / 2 + nreads * L * sp.sub_error_rate * k * 1.05
    per_rank = int(distinct / 0.5 / cores * 1.6) + 1024  # load <= 0.5 with room for the minimizer partition's imbalance
    o = O.Oracle(k, nranks=cores, nthreads=cores, capacity_per_rank=per_rank)
    t0 = time.perf_counter()
    o.add_reads(b, q, offs, block_reads=250_000)
    o.finalize_unsorted()
    dt = time.perf_counter() - t0
    st = o.stats()
    o.close()
    return {"value": st["raw_kmers"] / dt, "unit": "k-mers/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "rehashes": st["rehashes"],
            "sample": "%d reads x %d bp of the same synthetic stream with genomes scaled to the same depth as the GPU run, k=%d, "
                      "%d emulated ranks on %d threads, %.1f s" % (nreads, L, k, cores, cores, dt)}


def end_to_end(kc, k, L, d_bases, d_quals, d_offs, nreads, stage_ms):
    """The stage with the reads starting in HOST memory (pinned), H2D included: what a host caller of kc_submit_reads /
    kc_submit_packed_reads gets.  ASCII bases + qualities (2 B/base) and the read cache's packed bytes (1 B/base,
    src/packed_reads.cpp:99-126).  PCIe-bound by a wide margin: reported beside `value`, never as it."""
    import numpy as np
    import torch
    raw = nreads * (L - k + 1)
    hb = d_bases.cpu().pin_memory()
    hq = d_quals.cpu().pin_memory()
    ho = d_offs.cpu().numpy().astype(np.uint64)
    out = {}
    # the packed bytes: base code 0-4 (ACGTN) | min(quality - 33, 31) << 3, made on the GPU piece by piece
    code = torch.full((256,), 4, dtype=torch.uint8, device=d_bases.device)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    d_packed = torch.empty_like(d_bases)
    step = 1 << 28
    for o in range(0, d_bases.numel(), step):
        bb = d_bases[o:o + step]
        d_packed[o:o + step] = code[bb.long()] | ((d_quals[o:o + step].to(torch.int16) - 33).clamp(max=31).to(torch.uint8) << 3)
    hp = d_packed.cpu().pin_memory()
    del d_packed
    for name, fn, nbytes in (("ascii", lambda: kc.submit_reads(hb.numpy(), hq.numpy(), ho), 2 * nreads * L),
                             ("packed", lambda: kc.submit_packed_reads(hp.numpy(), ho), nreads * L)):
        best = None
        for _ in range(2):
            kc.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            kc.finalize()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        # the copies alone (the same pinned bytes, two at a time like the library's host pipe), for what the overlap hides:
        # overlap = the part of the stage's own time (input in HBM) that disappeared behind the copies
        halves = [hb, hq] if name == "ascii" else [hp[:hp.numel() // 2], hp[hp.numel() // 2:]]
        streams = [torch.cuda.Stream(device=d_bases.device) for _ in halves]
        dsts = [torch.empty_like(h, device=d_bases.device) for h in halves]
        copy_s = None
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for h, d, st_ in zip(halves, dsts, streams):
                with torch.cuda.stream(st_):
                    d.copy_(h, non_blocking=True)
            torch.cuda.synchronize()
            dtc = time.perf_counter() - t0
            copy_s = dtc if copy_s is None else min(copy_s, dtc)
        del dsts
        hidden_ms = max(0.0, copy_s * 1e3 + stage_ms - best * 1e3)
        out[name] = {"ms_per_step": best * 1e3, "value": raw / best, "unit": "k-mers/s", "input_bytes": nbytes,
                     "input_GBps_over_pcie": nbytes / best / 1e9, "copy_only_ms": copy_s * 1e3, "copy_only_GBps": nbytes / copy_s / 1e9,
                     "overlap": {"hidden_ms": hidden_ms, "of_stage_ms": stage_ms, "frac_of_stage": hidden_ms / stage_ms}}
    out["note"] = ("host-resident pinned input, H2D inside the timed region, results left in HBM; PCIe Gen5 x16 is 63 GB/s (spec); level 1 "
                   "and, in instalments, level 2 run behind the copies; the last block's share of level 2 and the count kernel come after "
                   "the last byte")
    return out


def main():
    a = parse_args()
    import torch
    import torch.distributed as dist
    import mhm2_kmer_analysis_v2_amd as pkg
    from mhm2_kmer_analysis_v2_amd.dist import ShardedKmerAnalysis

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d processes" % (a.gpus, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the only path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded_path = world > 1 or a.force_sharded
    if sharded_path:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    k, L = a.k, 150
    nl = pkg.lib().kc_num_longs(k)
    params = pkg.synth_params()
    nreads = a.reads
    raw_per_rank = nreads * (L - k + 1)

    # ---- input resident in HBM before any timed region
    d_bases = torch.empty(nreads * L, dtype=torch.uint8, device=dev)
    d_quals = torch.empty(nreads * L, dtype=torch.uint8, device=dev)
    d_offs = torch.empty(nreads + 1, dtype=torch.int64, device=dev)
    # distinct k-mers: genomes (shared by all ranks) + ~k per substitution error
    genome_kmers = 64 * 4_000_000
    est_unique = int((genome_kmers + world * nreads * L * params.sub_error_rate * k * 1.05) / world) + (1 << 20)
    ap_tuning = dict(mode=1) if a.table_path else ({k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(","))} if a.tune else None)
    # N>1: which exchange.  The single-pass flow costs one pass less on either side, but a shard then builds regions only
    # for the level-1 buckets it owns: kc_shard_capacity says whether those hold its k-mers (the same on every rank)
    flow = None
    if sharded_path:
        flow = "records" if (a.table_path or a.shard_flow == "records") else "single-pass"

    def make_counter(buckets):
        return pkg.KmerCounter(k, device=local_rank, rank_me=rank, rank_n=world, max_elems=est_unique, time_kernels=True,
                               max_kmers_buffered=int(nreads * (L - k - 1) * (1.05 if sharded_path else 1.02)) + (1 << 20), tuning=ap_tuning,
                               shard_buckets=buckets, wire_units=True)

    kc = make_counter(flow == "single-pass")
    if flow == "single-pass" and a.shard_flow == "auto":
        # decided together: a shard's capacity depends on how many buckets it owns, which differs between ranks when the
        # fan-out is no multiple of the shard count, and ranks on different flows would wait for each other for ever
        cap = torch.tensor([kc.shard_capacity()], dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(cap, op=dist.ReduceOp.MIN)
        if int(cap.item()) < est_unique:
            kc.close()
            flow = "records"
            kc = make_counter(False)
    # one explicit stream for everything (torch ops, RCCL enqueue order, the library's kernels): torch's default
    # stream is the null handle, which the library would read as "use your own stream"
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    kc.set_stream(stream.cuda_stream)
    kc.synth_reads_device(d_bases, d_quals, d_offs, nreads, first_read=rank * nreads, params=params)
    torch.cuda.synchronize()

    sharded = None
    if sharded_path:
        blk = min(a.block_reads, nreads)
        seg = int(blk * (L - k - 1) / world * 1.25) + 4096

        def extract(block, send, seg_cap):
            r0, r1 = block
            # offsets of a block must start at 0: shift a private copy
            offs = d_offs[r0:r1 + 1] - d_offs[r0]
            return kc.extract_partition(d_bases[r0 * L:], d_quals[r0 * L:], offs, send, seg_cap, nreads=r1 - r0)

        def shard_extract(block, send, seg_words):
            r0, r1 = block
            offs = d_offs[r0:r1 + 1] - d_offs[r0]
            return kc.shard_extract(d_bases[r0 * L:], d_quals[r0 * L:], offs, send, seg_words, nreads=r1 - r0)

        if flow == "single-pass":
            seg_words = (int(blk * (L - k - 1) / world * 1.1) * kc.rec_nl + 4096) if world > 1 else 1024
            sharded = ShardedKmerAnalysis.single_pass(kc, shard_extract, seg_words, dev)
        else:
            # units of the library's own wire record (kc_wire_unit): four six-byte records per three words at k = 21
            # (and a destination's records in `pieces` pieces by the top bits of their level-1 bucket: the receiver's level 1
            # then appends long runs to few buckets at a time)
            unit_words, unit_records, pieces = kc.wire_unit()
            sharded = ShardedKmerAnalysis(extract, lambda recv, n: kc.insert_records(recv, n), unit_words,
                                          int(seg / unit_records / pieces * 1.2) + 4096, dev, counter=kc, pieces=pieces,
                                          insert_pieces=kc.insert_record_pieces)

    def step():
        kc.reset()
        if not sharded_path:
            kc.submit_reads(d_bases, d_quals, d_offs, nreads=nreads)
        else:
            for r0 in range(0, nreads, a.block_reads):
                sharded.add_block((r0, min(nreads, r0 + a.block_reads)))
            sharded.finish()
        return kc.finalize()

    def fence():
        torch.cuda.synchronize()
        if sharded_path:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    kc.kernel_times(clear=True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ktimes = kc.kernel_times()
    st = kc.stats()

    checks = None
    if a.check:
        # size-independent properties at full size (the oracle cannot run 50 M reads in seconds)
        tot = torch.tensor([st["kmers_inserted"], st["raw_kmers"], st["total_kmers"], st["sum_counts"]], dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(tot)
        inserted, raw, total_kmers, sum_counts = (int(x) for x in tot.tolist())
        checks = {
            # every k-mer occurrence with two neighbours lands in exactly one shard's table
            "inserted_equals_expected": inserted == world * nreads * (L - k - 1),
            "raw_kmers_equals_expected": raw == world * raw_per_rank,
            "total_kmers": total_kmers, "sum_counts": sum_counts,
        }
        # idempotence: a second run over the same input gives the same set (checksum of checksums)
        def checksum(r):
            import numpy as np
            if int(r.n) == 0:
                return 0
            kk, cc, ll, rr = kc.results()  # host copies of the library-owned arrays
            h = (kk[:, 0] * np.uint64(0x9E3779B97F4A7C15)) ^ (cc.astype(np.uint64) << np.uint64(8)) \
                ^ ll.astype(np.uint64) ^ (rr.astype(np.uint64) << np.uint64(4))
            return int(np.bitwise_xor.reduce(h)) ^ (int(h.sum(dtype=np.uint64)) << 1)
        c1 = checksum(res)
        res2 = step()
        c2 = checksum(res2)
        checks["rerun_checksum_equal"] = c1 == c2

    value = world * raw_per_rank * a.steps / dt
    out = None
    if rank == 0:
        # Roofline of the STAGE: the contract's algorithmic bytes per raw k-mer x the raw k-mers one rank
        # puts through a step, over the step's wall time as timed above -- all kernels, gaps and host round trips in.
        # `kernels`: every kernel with ITS OWN algorithmic bytes (own_alg_bytes) over its own device time (HIP events
        # recorded on the launch stream inside the library), and its PMC traffic per step where a profile of this
        # workload is committed; the dominant one is named in `kernel`.
        tuned = bool(a.tune) or a.table_path or sharded_path
        traffic, traffic_src = pmc_traffic(nreads, k, tuned)
        step_s = dt / a.steps
        achieved = ALG_BYTES_PER_KMER[nl] * raw_per_rank / step_s / 1e9
        results_per_raw = st["total_kmers"] / float(raw_per_rank) if raw_per_rank else 0.0
        kernels = []
        for name, (launches, total_ms) in sorted(ktimes.items(), key=lambda kv: -kv[1][1]):
            ms_step = total_ms / a.steps
            own = own_alg_bytes(name, nl, k, L, results_per_raw)
            ent = {"name": name, "launches_per_step": launches / a.steps, "avg_launch_ms": total_ms / launches, "ms_per_step": ms_step}
            if own is not None and not sharded_path and not a.table_path:
                gbps = own * raw_per_rank / (ms_step * 1e-3) / 1e9
                ent.update({"alg_bytes_per_kmer": round(own, 3), "achieved_GBps": gbps, "frac": gbps / HBM_PEAK_GBPS})
            tr = next((v for n, v in traffic.items() if name in n), None)
            if tr is not None:
                ent.update({"traffic_bytes_per_step": tr["bytes"], "traffic_GBps": tr["bytes"] / (ms_step * 1e-3) / 1e9})
                if "l2_hit" in tr:
                    ent["l2_hit"] = round(tr["l2_hit"], 4)
            kernels.append(ent)
        dom = kernels[0] if kernels else None
        roof = {"bound": "hbm", "scope": "stage (all kernels of a step, wall time)", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "alg_bytes_per_kmer": ALG_BYTES_PER_KMER[nl],
                "kmers_per_step": raw_per_rank, "kernel": dom["name"] if dom else None,
                # (only the kernels that ran inside the timed steps: the profile also holds the set-up fills, the read
                # generator and the arena probe, which are no part of a step)
                "traffic": (sum(e["traffic_bytes_per_step"] for e in kernels if "traffic_bytes_per_step" in e) or None) if traffic else None,
                "traffic_unit": "bytes per step, summed over the kernels of a step",
                "l2_hit": {e["name"]: e["l2_hit"] for e in kernels if "l2_hit" in e} or None,
                "traffic_source": traffic_src, "kernels": kernels,
                "kernels_ms": {n: round(v[1] / a.steps, 3) for n, v in ktimes.items()}}
        out = {
            "metric": "k-mers/sec (kcount stage) at k=%d" % k, "value": value, "unit": "k-mers/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "ArcticSynth-shaped synthetic reads, %d reads x %d bp per GPU, k=%d, %s" % (
                nreads, L, k, "single-GPU hash table" if world == 1 else "%d shards, RCCL exchange" % world),
                "reads_per_gpu": nreads, "read_len": L, "k": k, "parallelism": "shard%d" % world, "shard_flow": flow},
            "roofline": roof,
            "results": {"total_kmers": st["total_kmers"], "num_unique": st["num_unique"], "capacity": st["capacity"],
                        "table_GB": st["table_bytes"] / 1e9},
        }
        if checks is not None:
            out["checks"] = checks
    # host-resident input (PCIe-inclusive) and the CPU baseline: rank 0 at N=1 only, outside the timed region
    if rank == 0 and world == 1 and not sharded_path and not a.no_end_to_end:
        out["end_to_end"] = end_to_end(kc, k, L, d_bases, d_quals, d_offs, nreads, dt / a.steps * 1e3)
    if rank == 0 and world == 1 and a.cpu_sample_reads > 0:
        kc.close()
        del d_bases, d_quals, d_offs
        out["cpu_baseline"] = cpu_baseline(k, min(a.cpu_sample_reads, nreads), nreads, params)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if sharded_path:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
