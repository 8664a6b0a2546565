"""The N>1 path on CPU: two processes over gloo run the same exchange code the GPU
ranks run over RCCL.  The device entry points (extract+bin, insert) are replaced by
CPU stand-ins built from the oracle's primitives and the library's host-callable
kc_owner, so what is under test is dist.py: counts exchange, all-to-all-v layout,
ownership, and that the union of the shards equals the single-rank oracle result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_records(reads, quals, k, world):
    """CPU stand-in for kc_extract_partition: records (canonical k-mer | ext codes) binned by kc_owner."""
    import mhm2_kmer_analysis_v2_amd as pkg
    from oracle import cpu_oracle as O
    L = pkg.lib()
    nl = O.num_longs(k)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    bins = [[] for _ in range(world)]
    for seq, ql in zip(reads, quals):
        if len(seq) < k + 2:
            continue
        kms = O.get_kmers(seq, k)
        for i in range(1, len(seq) - k):
            km = kms[i].copy()
            le = code.get(seq[i - 1], 4) if ord(ql[i - 1]) >= 33 + 20 else 4
            re = code.get(seq[i + k], 4) if ord(ql[i + k]) >= 33 + 20 else 4
            rc = O.revcomp(km, k)
            if tuple(rc) < tuple(km):
                km = rc
                le, re = (4 if re == 4 else 3 - re), (4 if le == 4 else 3 - le)
            owner = L.kc_owner(km.ctypes.data, k, world)
            rec = km.copy()
            rec[nl - 1] |= np.uint64(le | (re << 3))
            bins[owner].append(rec)
    return bins


def _worker(rank, world, port, k, tmp, flow="records"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import random_reads
        from mhm2_kmer_analysis_v2_amd.dist import ShardedKmerAnalysis
        from oracle import cpu_oracle as O
        nl = O.num_longs(k)
        rng = np.random.default_rng(77)  # same reads on every rank; each parses its own slice
        reads, quals = random_reads(rng, 240, min_len=k - 2, max_len=k + 80, genome_len=900)
        mine = list(range(rank, len(reads), world))
        table = {}

        def extract(block, send, seg):
            bins = make_records([reads[i] for i in block], [quals[i] for i in block], k, world)
            for d, recs in enumerate(bins):
                assert len(recs) <= seg
                if recs:
                    flat = np.concatenate(recs).astype(np.uint64).view(np.int64)
                    send[d * seg * nl:d * seg * nl + len(flat)] = torch.from_numpy(flat)
            return [len(b) for b in bins]

        # stand-in of the records flow in pieces (kc_wire_unit: a destination's records kept apart in Q pieces): piece
        # d * Q + q of the send buffer; what this rank keeps for itself is inserted piece by piece, what arrives comes
        # laid end to end
        Q = 4
        inserts = []

        def extract_pieces(block, send, seg):
            bins = make_records([reads[i] for i in block], [quals[i] for i in block], k, world)
            counts = [0] * (world * Q)
            for d, recs in enumerate(bins):
                for q in range(Q):
                    part = [r for r in recs if int(r[0] >> np.uint64(40)) % Q == q]
                    assert len(part) <= seg
                    if part:
                        flat = np.concatenate(part).astype(np.uint64).view(np.int64)
                        j = d * Q + q
                        send[j * seg * nl:j * seg * nl + len(flat)] = torch.from_numpy(flat)
                        counts[j] = len(part)
            return counts

        def insert(recv, n):
            inserts.append(n)
            a = recv[:n * nl].numpy().view(np.uint64).reshape(n, nl)
            for row in a:
                key = row.copy()
                le, re = int(row[nl - 1]) & 7, (int(row[nl - 1]) >> 3) & 7
                key[nl - 1] &= ~np.uint64(0x3F)
                e = table.setdefault(tuple(int(x) for x in key), [0, [0] * 4, [0] * 4])
                e[0] = min(e[0] + 1, 65535)
                if le < 4:
                    e[1][le] = min(e[1][le] + 1, 65535)
                if re < 4:
                    e[2][re] = min(e[2][re] + 1, 65535)

        # stand-ins of the single-pass flow (kc_shard_extract / kc_shard_reserve / kc_shard_commit): a wire segment is one
        # header word (its records) + the records; the rank's own share never enters a segment; what arrives is used
        # where reserve() put it
        reserved = []

        def shard_extract(block, send, seg_words):
            bins = make_records([reads[i] for i in block], [quals[i] for i in block], k, world)
            words = [0] * world
            for d, recs in enumerate(bins):
                if d == rank:
                    if recs:
                        insert(torch.from_numpy(np.concatenate(recs).astype(np.uint64).view(np.int64)), len(recs))
                    continue
                if not recs:
                    continue  # nothing for this shard: no segment at all
                flat = np.concatenate([np.array([len(recs)], dtype=np.uint64)] + recs).astype(np.uint64).view(np.int64)
                assert len(flat) <= seg_words
                send[d * seg_words:d * seg_words + len(flat)] = torch.from_numpy(flat)
                words[d] = len(flat)
            return words

        def shard_reserve(nwords):
            reserved.append(torch.full((nwords,), -1, dtype=torch.int64))
            return reserved[-1]

        def shard_commit(segment, nwords):
            assert any(segment.data_ptr() >= r.data_ptr() and segment.data_ptr() + 8 * nwords <= r.data_ptr() + 8 * r.numel() for r in reserved)
            assert segment.data_ptr() % 16 == 0
            n = int(segment[0])
            assert nwords == 1 + n * nl
            insert(segment[1:], n)

        if flow == "records":
            sk = ShardedKmerAnalysis(extract, insert, nl, seg_capacity=20000, device="cpu")
        elif flow == "records-pieces":
            sk = ShardedKmerAnalysis(extract_pieces, insert, nl, seg_capacity=6000, device="cpu", pieces=Q)
        else:
            sk = ShardedKmerAnalysis(shard_extract, shard_commit, 1, seg_capacity=20000 * nl + 1, device="cpu", reserve=shard_reserve)
        for b0 in range(0, len(mine), 50):  # several blocks, the last one ragged
            sk.add_block(mine[b0:b0 + 50])
        sk.add_block([])  # a rank with nothing to send still takes part
        sk.finish()
        # every record this rank holds is one it owns
        import mhm2_kmer_analysis_v2_amd as pkg
        L = pkg.lib()
        for key in list(table)[::7]:
            kw = np.array(key, dtype=np.uint64)
            assert L.kc_owner(kw.ctypes.data, k, world) == rank
        assert flow.startswith("records") or all(int((r == -1).sum()) <= 1 for r in reserved)  # every reserved word arrived (one pad word at most)
        tot = torch.tensor([sk.sent, sk.received], dtype=torch.int64)
        dist.all_reduce(tot)
        assert int(tot[0]) == int(tot[1])  # nothing lost or duplicated in flight
        # finalize this shard with the oracle's vote and write it out for the parent to merge
        out = []
        for key, (cnt, lc, rc_) in table.items():
            if cnt < 2:
                continue
            l, r = O.get_ext(lc, cnt), O.get_ext(rc_, cnt)
            if l in "XF" or r in "XF":
                continue
            out.append((key, cnt, l, r))
        torch.save(out, os.path.join(tmp, "shard%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("flow", ["records", "records-pieces", "single-pass"])
@pytest.mark.parametrize("k", [21, 51])
def test_two_rank_exchange_matches_single_rank_oracle(k, flow, tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, k, str(tmp_path), flow), nprocs=world, join=True)
    from helpers import random_reads, results_to_tuples
    from oracle import cpu_oracle as O
    rng = np.random.default_rng(77)
    reads, quals = random_reads(rng, 240, min_len=k - 2, max_len=k + 80, genome_len=900)
    (keys, counts, left, right), st = O.count_reads(reads, quals, k=k)
    want = results_to_tuples(keys, counts, left, right)
    got = []
    seen = set()
    for r in range(world):
        shard = torch.load(os.path.join(str(tmp_path), "shard%d.pt" % r), weights_only=False)
        for key, cnt, l, rr in shard:
            assert key not in seen  # no k-mer has two owners
            seen.add(key)
            got.append((key, cnt, l, rr))
    got.sort()
    assert got == want and len(want) > 20


def _failing_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mhm2_kmer_analysis_v2_amd.dist import ShardedKmerAnalysis

        def extract(block, send, seg):
            if rank == 1 and block == "bad":
                raise ValueError("segment too small")  # what KcError(KC_ERR_CAPACITY) is to the real extract
            return [0] * world

        sk = ShardedKmerAnalysis(extract, lambda recv, n: None, 1, seg_capacity=16, device="cpu")
        sk.add_block("fine")
        what = "no error"
        try:
            sk.add_block("bad")
        except ValueError as e:
            what = "own: %s" % e
        except RuntimeError as e:
            what = "peer: %s" % e
        with open(os.path.join(tmp, "rank%d.txt" % rank), "w") as f:
            f.write(what)
    finally:
        dist.destroy_process_group()


def test_a_failed_extraction_fails_every_rank_instead_of_hanging_the_others(tmp_path):
    """the rank whose extract raises still takes part in the exchange of the sizes, with sizes no block can have: it
    re-raises its own error, the others raise too (they used to wait in the collective for ever)"""
    world = 2
    mp.spawn(_failing_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read() for r in range(world)]
    assert got[1] == "own: segment too small"
    assert got[0].startswith("peer: rank(s) [1] failed")
