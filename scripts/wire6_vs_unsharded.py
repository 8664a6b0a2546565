"""The records flow with wire units (csrc/kc_wire6.hpp) against the unsharded pass on the same reads, entry by entry:
every k-mer the tables hold BEFORE the purge (kc_dump_table: singletons too), its count and its eight extension
counters.  R shards live on the one GPU; the blocks rotate over them as senders.
Usage: python scripts/wire6_vs_unsharded.py [reads=5000000] [R=2] [block=1000000]"""
import sys

import numpy as np
import torch

import mhm2_kmer_analysis_v2_amd as pkg

nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2
block = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
k, L = 21, 150
p = pkg.synth_params()
db = torch.empty(nreads * L, dtype=torch.uint8, device="cuda")
dq = torch.empty(nreads * L, dtype=torch.uint8, device="cuda")
do = torch.empty(nreads + 1, dtype=torch.int64, device="cuda")
occ = nreads * (L - k - 1)
est = int(64 * 4_000_000 + nreads * L * 0.005 * k * 1.05) + (1 << 20)
with pkg.KmerCounter(k, max_elems=est, max_kmers_buffered=int(occ * 1.02) + (1 << 20)) as kc:
    kc.synth_reads_device(db, dq, do, nreads, params=p)
    kc.submit_reads(db, dq, do, nreads=nreads)
    kc.flush()
    want = kc.dump_table()
    print("unsharded: %d entries, %d occurrences" % (len(want[1]), int(want[1].astype(np.uint64).sum())), flush=True)

shards = [pkg.KmerCounter(k, rank_me=r, rank_n=R, max_elems=est // R + (1 << 20), max_kmers_buffered=int(occ / R * 1.1) + (1 << 20), wire_units=True,
                          tuning=dict(p1=1024, p2=1024))  # (the benchmark's fan-outs: six-byte records whatever the size of this run)
          for r in range(R)]
uw, ur, Q = shards[0].wire_unit()
assert (uw, ur) == (3, 4), (uw, ur)
seg = int(block * (L - k - 1) / R / Q * 1.3) // ur + 8192  # units per piece
send = torch.zeros(R * Q * seg * uw, dtype=torch.int64, device="cuda")
units = 0
for i, r0 in enumerate(range(0, nreads, block)):
    r1 = min(nreads, r0 + block)
    offs = do[r0:r1 + 1] - do[r0]
    counts = shards[i % R].extract_partition(db[r0 * L:], dq[r0 * L:], offs, send, seg, nreads=r1 - r0)
    units += int(counts.sum())
    for d in range(R):  # a destination's pieces laid end to end, as an exchange lays what it receives
        flat = torch.cat([send[(d * Q + q) * seg * uw:(d * Q + q) * seg * uw + int(counts[d * Q + q]) * uw] for q in range(Q)])
        shards[d].insert_records(flat, flat.numel() // uw)
        shards[d].flush()
print("shipped %d units = %d slots for %d occurrences" % (units, units * ur, occ), flush=True)
parts = []
for r, s in enumerate(shards):
    t = s.dump_table()
    own = np.array([s.partition_owner(t[0][i]) for i in range(0, len(t[1]), max(1, len(t[1]) // 200))])
    assert (own == r).all()
    parts.append(t)
    print("shard %d: %d entries, inserted %d" % (r, len(t[1]), s.stats()["kmers_inserted"]), flush=True)
    s.close()
keys = np.concatenate([t[0] for t in parts])
order = np.argsort(keys[:, 0], kind="stable")
got = tuple(np.concatenate([t[i] for t in parts])[order] for i in range(3))
ok = len(got[1]) == len(want[1]) and (got[0] == want[0]).all() and (got[1] == want[1]).all() and (got[2] == want[2]).all()
if not ok:
    a, b = want[0][:, 0], got[0][:, 0]
    only_w, only_g = np.setdiff1d(a, b), np.setdiff1d(b, a)
    print("entries: want %d got %d; only unsharded %d, only shards %d" % (len(a), len(b), len(only_w), len(only_g)))
    for x in only_w[:10]:
        i = np.searchsorted(a, x)
        print("  missing  %016x count %d ext %s" % (int(x), int(want[1][i]), want[2][i].tolist()))
    for x in only_g[:10]:
        i = np.searchsorted(b, x)
        print("  phantom  %016x count %d ext %s" % (int(x), int(got[1][i]), got[2][i].tolist()))
    if not len(only_w) and not len(only_g):
        bad = np.nonzero((got[1] != want[1]) | (got[2] != want[2]).any(axis=1))[0]
        print("same keys; %d entries differ" % len(bad))
        for i in bad[:10]:
            print("  %016x want %d %s got %d %s" % (int(a[i]), int(want[1][i]), want[2][i].tolist(), int(got[1][i]), got[2][i].tolist()))
    sys.exit(1)
print("identical: %d entries" % len(want[1]))
