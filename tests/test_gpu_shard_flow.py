"""The single-pass shard flow (kc_shard_extract / kc_shard_reserve / kc_shard_commit, csrc/kc_shard.hpp) with several
shards living on one device: the union of the shards' results is bit-exact the oracle's, every k-mer sits on the shard
kc_shard_owner names, nothing is counted twice or lost.  Replaces kmer_dht.cpp:143-151,247-258 + gpu_hash_table.cpp:655-695."""
import numpy as np
import pytest

import mhm2_kmer_analysis_v2_amd as pkg
from helpers import random_reads
from oracle import cpu_oracle as O
from test_gpu_parity import PATHS, arrays, assert_same, oracle_run

pytestmark = pytest.mark.gpu


def run_shards(reads, quals, k, R, tuning, blocks=2, max_kmers_buffered=0, time_kernels=False):
    """Every 'rank' extracts its slice of the reads in `blocks` blocks; segments travel by a device copy."""
    import torch
    shards = [pkg.KmerCounter(k, rank_me=r, rank_n=R, tuning=tuning, max_kmers_buffered=max_kmers_buffered, time_kernels=time_kernels)
              for r in range(R)]
    nl = shards[0].rec_nl
    total = sum(max(0, len(r) - k - 1) for r in reads)
    seg_words = total * nl + 2048
    segs = torch.zeros(R * seg_words, dtype=torch.int64, device="cuda")
    per = (len(reads) + R - 1) // R
    shipped = 0
    run_shards.records = run_shards.segments = 0  # (of the last call: foreign records shipped and the segments they travelled in)
    for blk in range(blocks):
        for r in range(R):
            mine = list(range(r * per, min(len(reads), (r + 1) * per)))
            part = mine[blk * len(mine) // blocks:(blk + 1) * len(mine) // blocks]
            bb, qq, oo = arrays([reads[i] for i in part], [quals[i] for i in part])
            words = shards[r].shard_extract(bb, qq, oo, segs, seg_words)
            assert int(words[r]) == 0
            for d in range(R):
                w = int(words[d])
                if d == r or not w:
                    continue
                dst = shards[d].shard_reserve(w)
                dst.copy_(segs[d * seg_words:d * seg_words + w])
                torch.cuda.synchronize()  # the copy ran on torch's stream, the contexts have streams of their own
                shards[d].shard_commit(dst, w)
                shipped += w
                run_shards.records += int(dst[2].item()) + (int(dst[1].item()) >> 32)  # header: records, loose records
                run_shards.segments += 1
    return shards, shipped, total


def union(parts):
    keys = np.concatenate([p[0] for p in parts])
    order = np.lexsort([keys[:, j] for j in range(keys.shape[1] - 1, -1, -1)])
    return tuple(np.concatenate([p[i] for p in parts])[order] for i in range(4))


CASES = [
    (21, 2, "compact"), (21, 3, "compact"), (21, 4, "compact-short"), (21, 3, "bucketed-odd"), (21, 2, "wide"), (21, 3, "bucketed"),
    (17, 3, "compact"), (31, 3, "bucketed-small"), (51, 2, "bucketed"), (51, 3, "bucketed-odd"), (77, 4, "bucketed-small"), (99, 2, "bucketed"),
]


@pytest.mark.parametrize("k,R,path", CASES)
def test_shard_flow_matches_oracle(k, R, path):
    rng = np.random.default_rng(100 + k + R)
    reads, quals = random_reads(rng, 1200, min_len=30, max_len=150, genome_len=3000)
    b, q, offs = arrays(reads, quals)
    want, _, wst = oracle_run(b, q, offs, k)
    shards, shipped, total = run_shards(reads, quals, k, R, PATHS[path])
    assert shipped > 0
    # words on the wire per foreign record: five bytes in the compact wire form (short-form compact records: k=21 with
    # 1024 level-1 buckets), the record's words otherwise -- plus header and padding per segment and bucket
    P1 = (PATHS[path] or {}).get("p1", 0)
    if path == "compact-short":
        assert shipped * 8 <= 5 * run_shards.records + run_shards.segments * 8 * (4 + P1 // 2 + 2 * P1 + 4)
    parts = [s.sorted_results() for s in shards]
    assert_same(union(parts), want)
    st = [s.stats() for s in shards]
    assert sum(x["kmers_inserted"] for x in st) == wst["kmers_inserted"] == total
    assert sum(x["num_unique"] for x in st) == wst["unique"] and sum(x["num_purged"] for x in st) == wst["purged"]
    for r, p in enumerate(parts):
        assert len(p[1]) > 0
        for i in range(0, len(p[1]), 29):
            assert shards[r].shard_owner(p[0][i]) == r
    for s in shards:
        s.close()


@pytest.mark.parametrize("k", [21, 51])
@pytest.mark.parametrize("tuning", [
    dict(writers=2, p1=4, p2=4, slots=4096, chunk1=16, chain1_max=8, ovf_capacity=1 << 20),   # level-1 chains overflow: loose records
    dict(writers=2, p1=4, p2=4, slots=4096, chunk1=16, arena1=24, ovf_capacity=1 << 20),      # a writer's arena runs out
    dict(writers=4, p1=8, p2=8, slots=128, chunk1=16, chain1_max=6, chunk2=16, chain2_max=12, ovf_capacity=1 << 20),  # all at once
], ids=["chain1-overflow", "arena1-exhausted", "everything"])
def test_shard_flow_overflow_records_travel_loose(k, tuning):
    rng = np.random.default_rng(7 + k)
    reads, quals = random_reads(rng, 1200, min_len=40, max_len=160, genome_len=4000)
    b, q, offs = arrays(reads, quals)
    want, _, wst = oracle_run(b, q, offs, k)
    shards, _, total = run_shards(reads, quals, k, 2, tuning, blocks=3)
    parts = [s.sorted_results() for s in shards]
    assert_same(union(parts), want)
    assert sum(s.stats()["kmers_inserted"] for s in shards) == total
    for s in shards:
        s.close()


@pytest.mark.parametrize("k,R", [(21, 2), (21, 4), (51, 3)])
def test_many_small_blocks_do_not_use_the_arena_up(k, R):
    """Every block empties the chains of the buckets other shards own.  Their chunks come from the top of the writers'
    arenas and the top is given back after every block (ChainDest::own_lo, kc_shard_release_kernel): forty blocks through
    arenas with room for little more than the reads themselves must neither overflow nor spill a record to the global
    table (a bump allocator that only grew lost half a chunk per foreign chain and block)."""
    rng = np.random.default_rng(300 + k + R)
    reads, quals = random_reads(rng, 2400, min_len=60, max_len=150, genome_len=5000)
    b, q, offs = arrays(reads, quals)
    want, _, wst = oracle_run(b, q, offs, k)
    total = sum(max(0, len(r) - k - 1) for r in reads)
    # per writer: the share of one shard's reads in 16-record chunks, plus one open chunk per bucket and a little slack
    W, P = 2, 16
    arena1 = (total // R) // (W * 16) + 2 * P + 8
    tuning = dict(writers=W, p1=P, p2=16, slots=1024, chunk1=16, arena1=arena1, ovf_capacity=1 << 16)
    shards, shipped, _ = run_shards(reads, quals, k, R, tuning, blocks=40, time_kernels=True)
    assert shipped > 0
    parts = [s.sorted_results() for s in shards]
    assert_same(union(parts), want)
    st = [s.stats() for s in shards]
    assert sum(x["kmers_inserted"] for x in st) == total
    # nothing took the fallback kernels (overflow records and flagged regions to the global table): the fast path held
    # all forty blocks
    for s in shards:
        assert "kc_flagged_to_table_kernel" not in s.kernel_times(), s.kernel_times()
        s.close()


def test_single_shard_is_the_plain_flow():
    k = 21
    rng = np.random.default_rng(5)
    reads, quals = random_reads(rng, 600, min_len=30, max_len=150, genome_len=2000)
    b, q, offs = arrays(reads, quals)
    want, _, _ = oracle_run(b, q, offs, k)
    with pkg.KmerCounter(k) as kc:
        words = kc.shard_extract(b, q, offs, None, 0)
        assert int(words[0]) == 0
        assert_same(kc.sorted_results(), want)


def test_shard_flow_errors():
    import torch
    k = 21
    rng = np.random.default_rng(6)
    reads, quals = random_reads(rng, 600, min_len=30, max_len=150, genome_len=2000)
    b, q, offs = arrays(reads, quals)
    segs = torch.zeros(2 * 4096, dtype=torch.int64, device="cuda")
    with pkg.KmerCounter(k, rank_me=0, rank_n=2, tuning=PATHS["compact"]) as kc:
        with pytest.raises(pkg.KcError) as e:  # a segment too small is reported, nothing is shipped or lost
            kc.shard_extract(b, q, offs, segs, 4096)
        assert e.value.status == -6
        with pytest.raises(pkg.KcError) as e:  # the two ownership rules do not mix in one pass
            kc.submit_reads(b, q, offs)
        assert e.value.status == -8
        with pytest.raises(pkg.KcError):  # memory that is not the context's
            kc.shard_commit(segs, 64)
    with pkg.KmerCounter(k, rank_me=0, rank_n=2, tuning=dict(mode=1)) as kc:  # the global-table path has no buckets to own
        with pytest.raises(pkg.KcError) as e:
            kc.shard_extract(b, q, offs, segs, 4096)
        assert e.value.status == -8
    # a segment made for another geometry is refused
    a = pkg.KmerCounter(k, rank_me=0, rank_n=2, tuning=PATHS["compact"])
    c = pkg.KmerCounter(k, rank_me=1, rank_n=2, tuning=PATHS["compact-short"])
    big = torch.zeros(2 * (1 << 18), dtype=torch.int64, device="cuda")
    words = a.shard_extract(b, q, offs, big, 1 << 18)
    w = int(words[1])
    dst = c.shard_reserve(w)
    dst.copy_(big[(1 << 18):(1 << 18) + w])
    torch.cuda.synchronize()
    with pytest.raises(pkg.KcError) as e:
        c.shard_commit(dst, w)
    assert e.value.status == -1
    a.close()
    c.close()
    # a segment whose bucket counts do not add up to its header's record count is refused, not read past its end
    a = pkg.KmerCounter(k, rank_me=0, rank_n=2, tuning=PATHS["compact"])
    c = pkg.KmerCounter(k, rank_me=1, rank_n=2, tuning=PATHS["compact"])
    words = a.shard_extract(b, q, offs, big, 1 << 18)
    w = int(words[1])
    dst = c.shard_reserve(w)
    dst.copy_(big[(1 << 18):(1 << 18) + w])
    dst[4] += 1  # the first two u32 counts live in word 4 (behind the four header words)
    torch.cuda.synchronize()
    with pytest.raises(pkg.KcError) as e:
        c.shard_commit(dst, w)
    assert e.value.status == -1 and b"add up" in pkg.lib().kc_last_error()
    a.close()
    c.close()


def test_shard_flow_again_after_reset():
    """kc_reset starts a new pass: the received segments are forgotten, their memory is reused, the result is the same."""
    import torch
    k, R = 21, 2
    rng = np.random.default_rng(17)
    reads, quals = random_reads(rng, 800, min_len=30, max_len=150, genome_len=2500)
    b, q, offs = arrays(reads, quals)
    want, _, _ = oracle_run(b, q, offs, k)
    shards = [pkg.KmerCounter(k, rank_me=r, rank_n=R, tuning=PATHS["compact"]) for r in range(R)]
    total = sum(max(0, len(r) - k - 1) for r in reads)
    seg_words = total + 2048
    segs = torch.zeros(R * seg_words, dtype=torch.int64, device="cuda")
    first_ptrs = None
    for rep in range(3):
        ptrs = []
        for r in range(R):
            part = list(range(r, len(reads), R))
            bb, qq, oo = arrays([reads[i] for i in part], [quals[i] for i in part])
            words = shards[r].shard_extract(bb, qq, oo, segs, seg_words)
            d = 1 - r
            w = int(words[d])
            dst = shards[d].shard_reserve(w)
            ptrs.append(dst.data_ptr())
            dst.copy_(segs[d * seg_words:d * seg_words + w])
            torch.cuda.synchronize()
            shards[d].shard_commit(dst, w)
        assert_same(union([s.sorted_results() for s in shards]), want)
        if first_ptrs is None:
            first_ptrs = ptrs
        assert ptrs == first_ptrs  # the extents of the first pass are reused
        for s in shards:
            s.reset()
    for s in shards:
        s.close()


def test_too_many_segments_in_one_pass_is_reported():
    import torch
    k = 21
    tuning = dict(writers=300, p1=256, p2=256, slots=512)  # 512 - 300 writers leave room for 212 flat sources
    a = pkg.KmerCounter(k, rank_me=0, rank_n=2, tuning=tuning)
    c = pkg.KmerCounter(k, rank_me=1, rank_n=2, tuning=tuning)
    reads, quals = ["ACGTTGCATGCCGTAAGCTTAGCGATCGATTGCA" * 2], ["I" * 68]
    b, q, offs = arrays(reads, quals)
    segs = torch.zeros(2 * 4096, dtype=torch.int64, device="cuda")
    with pytest.raises(pkg.KcError) as e:
        for i in range(300):
            words = a.shard_extract(b, q, offs, segs, 4096)
            w = int(words[1])
            assert w > 0
            dst = c.shard_reserve(w)
            dst.copy_(segs[4096:4096 + w])
            torch.cuda.synchronize()
            c.shard_commit(dst, w)
    assert e.value.status == -6 and i == 212
    a.close()
    c.close()


def test_shard_capacity_follows_the_geometry():
    """kc_shard_capacity: a shard's share of the regions; KC_FLAG_SHARD_BUCKETS sizes the geometry for the flow."""
    k, est = 21, 40_000_000
    with pkg.KmerCounter(k, rank_me=0, rank_n=1, max_elems=est) as one, \
            pkg.KmerCounter(k, rank_me=1, rank_n=4, max_elems=est) as plain, \
            pkg.KmerCounter(k, rank_me=1, rank_n=4, max_elems=est, shard_buckets=True) as flow:
        c1, cp, cf = one.shard_capacity(), plain.shard_capacity(), flow.shard_capacity()
    assert c1 >= est                 # a single shard owns every region of a geometry made for est
    assert cp * 3 < c1               # a quarter of the same geometry
    assert cf >= est                 # the flag makes the geometry four times as large: the quarter holds est again


@pytest.mark.parametrize("k,R", [(21, 2), (51, 3)])
def test_shard_flow_from_seq_blocks(k, R):
    """kc_shard_extract_seq_block: the reference's '_'-joined case-masked block as the sender's input."""
    import torch
    rng = np.random.default_rng(61 + k)
    reads, quals = random_reads(rng, 900, min_len=30, max_len=150, genome_len=2500)
    b, q, offs = arrays(reads, quals)
    want, _, _ = oracle_run(b, q, offs, k)
    masked = ["".join(c.lower() if ord(x) < 33 + 20 else c for c, x in zip(r, ql)) for r, ql in zip(reads, quals)]
    shards = [pkg.KmerCounter(k, rank_me=r, rank_n=R, tuning=PATHS["compact"] if k == 21 else None) for r in range(R)]
    nl = shards[0].rec_nl
    seg_words = sum(max(0, len(r) - k - 1) for r in reads) * nl + 2048
    segs = torch.zeros(R * seg_words, dtype=torch.int64, device="cuda")
    for r in range(R):
        for piece in range(2):
            part = masked[r::R][piece::2]
            words = shards[r].shard_extract_seq_block("_".join(part).encode(), segs, seg_words)
            for d in range(R):
                w = int(words[d])
                if d == r or not w:
                    continue
                dst = shards[d].shard_reserve(w)
                dst.copy_(segs[d * seg_words:d * seg_words + w])
                torch.cuda.synchronize()
                shards[d].shard_commit(dst, w)
    assert_same(union([s.sorted_results() for s in shards]), want)
    for s in shards:
        s.close()
