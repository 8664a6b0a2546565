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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=50_000_000, help="reads per GPU (150 bp each)")
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--block-reads", type=int, default=8_000_000, help="N>1: reads per exchange block")
    ap.add_argument("--cpu-sample-reads", type=int, default=1_000_000, help="reads timed through the CPU oracle (0 = skip)")
    ap.add_argument("--check", action="store_true", help="size-independent result checks after the timed region")
    ap.add_argument("--table-path", action="store_true", help="A/B: force the global-table insert path instead of the bucketed one")
    ap.add_argument("--force-sharded", action="store_true", help="run the N>1 code path (bin by owner, exchange, insert records) even at N=1")
    ap.add_argument("--tune", default="", help="A/B: bucketed-path geometry overrides, e.g. slots=2048,p1=1024,p2=1024")
    return ap.parse_args()


def pmc_traffic(kernel_name, nreads, k, tuned):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes of this same workload
    (scripts/pmc_profile.sh: FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950).  None when no profile matches."""
    if nreads != 50_000_000 or k != 21 or tuned:
        return None, None
    path = os.path.join(ROOT, "profiles", "r01_bucketed_final_pmc_50Mreads.json")
    try:
        prof = json.load(open(path))
    except Exception:
        return None, None
    for name, ctr in prof.items():
        if kernel_name in name and "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
            return (2.0 * ctr["FETCH_SIZE"] + ctr["WRITE_SIZE"]) * 1024.0, os.path.relpath(path, ROOT)
    return None, None


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


def cpu_baseline(k, nreads, params):
    """The oracle (a port of the reference CPU kcount, oracle/kcount_oracle.c) on this host's cores,
    on a bounded sample of the same read stream.  Reported beside the GPU number; never the target."""
    import mhm2_kmer_analysis_v2_amd as pkg
    from oracle import cpu_oracle as O
    cores = usable_cores()
    b, q, offs = pkg.synth_reads_host(nreads, params=params)
    raw = nreads * (params.read_len - k + 1)  # fixed-length reads
    # table sized by the reference's rule so that it never grows inside the timed region:
    # (adjusted + errors) / 0.66 with sequencing_depth 4, BASE_ERROR_RATE 0.005 (kmer_dht.cpp:126-131)
    per_rank = int((raw / 4 + raw * (1 - (1 - 0.005) ** k)) / 0.66 / cores) + 1024
    o = O.Oracle(k, nranks=cores, nthreads=cores, capacity_per_rank=per_rank)
    t0 = time.perf_counter()
    o.add_reads(b, q, offs, block_reads=250_000)
    o.finalize()
    dt = time.perf_counter() - t0
    st = o.stats()
    o.close()
    return {"value": st["raw_kmers"] / dt, "unit": "k-mers/s", "cores": cores, "kind": "port",
            "sample": "%d reads x %d bp of the same synthetic stream, k=%d, %d emulated ranks, %.1f s" % (
                nreads, params.read_len, k, cores, dt)}


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
    kc = pkg.KmerCounter(k, device=local_rank, rank_me=rank, rank_n=world, max_elems=est_unique, time_kernels=True,
                         max_kmers_buffered=int(nreads * (L - k - 1) * 1.02) + (1 << 20), tuning=ap_tuning)
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

        sharded = ShardedKmerAnalysis(extract, lambda recv, n: kc.insert_records(recv, n), nl, seg, dev)

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
        # dominant kernel by device time; HIP events recorded on the launch stream inside the library
        dom = max(ktimes.items(), key=lambda kv: kv[1][1]) if ktimes else None
        roof = None
        if dom:
            name, (launches, total_ms) = dom
            avg_ms = total_ms / launches
            # algorithmic bytes per launch = contract bytes per raw k-mer x raw k-mers one launch processes
            raw_per_launch = raw_per_rank * a.steps / launches
            achieved = ALG_BYTES_PER_KMER[nl] * raw_per_launch / (avg_ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(name, nreads, k, bool(a.tune) or a.table_path or sharded_path)
            roof = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                    "launches": launches, "avg_launch_ms": avg_ms,
                    "alg_bytes_per_kmer": ALG_BYTES_PER_KMER[nl], "kmers_per_launch": raw_per_launch,
                    "kernels_ms": {n: round(v[1] / a.steps, 3) for n, v in ktimes.items()}}
        out = {
            "metric": "k-mers/sec (kcount stage) at k=%d" % k, "value": value, "unit": "k-mers/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "ArcticSynth-shaped synthetic reads, %d reads x %d bp per GPU, k=%d, %s" % (
                nreads, L, k, "single-GPU hash table" if world == 1 else "%d shards, RCCL exchange" % world),
                "reads_per_gpu": nreads, "read_len": L, "k": k, "parallelism": "shard%d" % world},
            "roofline": roof,
            "results": {"total_kmers": st["total_kmers"], "num_unique": st["num_unique"], "capacity": st["capacity"],
                        "table_GB": st["table_bytes"] / 1e9},
        }
        if checks is not None:
            out["checks"] = checks
    # CPU baseline: rank 0 at N=1 only, bounded sample
    if rank == 0 and world == 1 and a.cpu_sample_reads > 0:
        kc.close()
        del d_bases, d_quals, d_offs
        out["cpu_baseline"] = cpu_baseline(k, min(a.cpu_sample_reads, nreads), params)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if sharded_path:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
