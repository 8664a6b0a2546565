"""Randomised parity sweep on the GPU against the oracle: random k, geometry, read shapes.
python scripts/stress_parity.py [cases] [seed] [scale]
KC_STRESS_KS=21,51 restricts the k, KC_STRESS_FLOW=records the flow (plain, small-buffer, shards, records)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # before the library touches the device
import mhm2_kmer_analysis_v2_amd as pkg
from oracle import cpu_oracle as O
from helpers import random_reads

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # multiplies the number of reads and the genome length
rng = np.random.default_rng(seed)
KS = [11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 30, 31, 33, 47, 51, 61, 62, 63, 65, 77, 93, 95]
if os.environ.get("KC_STRESS_KS"):
    KS = [int(x) for x in os.environ["KC_STRESS_KS"].split(",")]
bad = 0
for c in range(cases):
    k = int(rng.choice(KS))
    nreads = int(rng.integers(200, 6000)) * scale
    genome = int(rng.integers(500, 60000)) * scale
    reads, quals = random_reads(rng, nreads, min_len=int(rng.integers(1, k + 3)), max_len=k + int(rng.integers(2, 200)),
                                genome_len=genome, err=float(rng.choice([0.0, 0.005, 0.03])), n_rate=float(rng.choice([0.0, 0.01, 0.1])))
    b, q, offs = O.reads_to_arrays(reads, quals)
    o = O.Oracle(k, nranks=int(rng.integers(1, 5)), nthreads=8)
    o.add_reads(b, q, offs)
    want = o.finalize()
    o.close()
    la, lb = int(rng.integers(1, 11)), int(rng.integers(1, 11))
    tuning = rng.choice(["auto", "pow2", "odd", "tiny-chunks", "wide"])
    t = None
    if tuning == "pow2":
        t = dict(writers=int(rng.integers(1, 9)), p1=1 << la, p2=1 << lb, slots=int(rng.choice([256, 1024, 2048, 4096])))
    elif tuning == "odd":
        t = dict(writers=int(rng.integers(1, 9)), p1=int(rng.integers(1, 300)), p2=int(rng.integers(1, 300)), slots=int(rng.choice([128, 512, 2048])))
    elif tuning == "tiny-chunks":
        t = dict(writers=int(rng.integers(1, 5)), p1=1 << min(la, 6), p2=1 << min(lb, 6), slots=2048, chunk1=16, chunk2=16,
                 chain1_max=int(rng.integers(4, 40)), chain2_max=int(rng.integers(4, 40)), ovf_capacity=1 << 20)
    elif tuning == "wide":
        t = dict(mode=2, p1=1 << min(la, 8), p2=1 << min(lb, 8))
    occ = sum(max(0, len(r) - k - 1) for r in reads)
    flow = rng.choice(["plain", "plain", "small-buffer", "shards", "records"])
    flow = os.environ.get("KC_STRESS_FLOW", flow)
    if flow == "small-buffer" and t is not None:
        t["ovf_capacity"] = 1 << 20  # (the lists of a tiny buffer would not hold one tile)
    try:
        if flow == "shards":
            R = int(rng.integers(2, 5))
            if t is not None and t.get("p1", 1024) < R:
                t["p1"] = 8
            shards = [pkg.KmerCounter(k, rank_me=r, rank_n=R, max_kmers_buffered=(1 << 22) * scale, tuning=t) for r in range(R)]
            nl = shards[0].rec_nl
            seg_words = occ * nl + 4096
            segs = torch.zeros(R * seg_words, dtype=torch.int64, device="cuda")
            nb = int(rng.integers(1, 4))
            for r in range(R):
                mine = list(range(r, nreads, R))
                for piece in range(nb):
                    part = mine[piece * len(mine) // nb:(piece + 1) * len(mine) // nb]
                    bb, qq, oo = O.reads_to_arrays([reads[i] for i in part], [quals[i] for i in part])
                    words = shards[r].shard_extract(bb, qq, oo, segs, seg_words)
                    for d in range(R):
                        w = int(words[d])
                        if d != r and w:
                            dst = shards[d].shard_reserve(w)
                            dst.copy_(segs[d * seg_words:d * seg_words + w])
                            torch.cuda.synchronize()
                            shards[d].shard_commit(dst, w)
            parts = [sh.sorted_results() for sh in shards]
            for sh in shards:
                sh.close()
            keys = np.concatenate([p[0] for p in parts])
            order = np.lexsort([keys[:, j] for j in range(keys.shape[1] - 1, -1, -1)])
            got = tuple(np.concatenate([p[i] for p in parts])[order] for i in range(4))
        elif flow == "records":  # kc_extract_partition / kc_insert_records, k-mer records or wire units (csrc/kc_wire6.hpp)
            R = int(rng.integers(1, 6))
            wu = bool(rng.integers(0, 2))
            if wu and k == 21 and rng.integers(0, 2):
                t = dict(writers=int(rng.integers(1, 9)), p1=1024, p2=1 << int(rng.integers(6, 11)), slots=int(rng.choice([256, 1024, 2048])))  # six-byte records
            small = bool(rng.integers(0, 2))
            cap = max(40000, int(occ / R * 0.5)) if small else (1 << 22) * scale
            if small and t is not None:
                t["ovf_capacity"] = 1 << 20
            shards = [pkg.KmerCounter(k, rank_me=r, rank_n=R, max_kmers_buffered=cap, tuning=t, wire_units=wu) for r in range(R)]
            uw, ur, Q = shards[0].wire_unit()
            flow = "records(%d shards, unit %d words / %d records, %d pieces%s)" % (R, uw, ur, Q, ", small buffer" if small else "")
            nb = int(rng.integers(2, 7))
            seg = occ // ur + 8192
            recs = torch.zeros(R * Q * seg * uw, dtype=torch.int64, device="cuda")
            for piece in range(nb):
                part = list(range(piece * nreads // nb, (piece + 1) * nreads // nb))
                if not part:
                    continue
                bb, qq, oo = O.reads_to_arrays([reads[i] for i in part], [quals[i] for i in part])
                counts = shards[piece % R].extract_partition(bb, qq, oo, recs, seg)
                for j in range(R * Q):
                    if int(counts[j]):
                        shards[j // Q].insert_records(recs[j * seg * uw:], int(counts[j]))
                for d in range(R):
                    shards[d].flush()
            parts = [sh.sorted_results() for sh in shards]
            for sh in shards:
                sh.close()
            keys = np.concatenate([p[0] for p in parts])
            order = np.lexsort([keys[:, j] for j in range(keys.shape[1] - 1, -1, -1)])
            got = tuple(np.concatenate([p[i] for p in parts])[order] for i in range(4))
        else:
            cap = (1 << 22) * scale if flow == "plain" else max(40000, int(occ * float(rng.choice([0.3, 0.6, 1.5]))))
            with pkg.KmerCounter(k, max_kmers_buffered=cap, tuning=t) as kc:
                nb = int(rng.integers(1, 4)) if flow == "plain" else int(rng.integers(3, 9))  # submit in several pieces
                cuts = sorted(set([0, nreads] + [int(x) for x in rng.integers(0, nreads + 1, size=nb - 1)]))
                for a0, a1 in zip(cuts[:-1], cuts[1:]):
                    bb, qq, oo = O.reads_to_arrays(reads[a0:a1], quals[a0:a1])
                    if a1 > a0:
                        kc.submit_reads(bb, qq, oo)
                got = kc.sorted_results()
        ok = all(g.shape == w.shape and (g == w).all() for g, w in zip(got, want))
    except Exception as e:  # noqa
        ok = False
        print("  exception:", repr(e))
    if not ok:
        bad += 1
    print("case %d k=%d reads=%d genome=%d %s tuning=%s %s -> %s (%d k-mers)" % (c, k, nreads, genome, flow, tuning, t, "ok" if ok else "MISMATCH", len(want[1])), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
