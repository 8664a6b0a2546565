"""Parity tests proper: the HIP path through the C ABI against the oracle, bit-exact
(integer / byte / index work: no tolerance).  Need a real MI355X."""
import numpy as np
import pytest

import mhm2_kmer_analysis_v2_amd as pkg
from helpers import random_reads
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu


def oracle_run(bases, quals, offs, k, nranks=3, nthreads=2, dmin_thres=2):
    o = O.Oracle(k, nranks=nranks, nthreads=nthreads, dmin_thres=dmin_thres)
    o.add_reads(bases, quals, offs)
    table = o.dump_table()
    res = o.finalize()
    st = o.stats()
    assert st["dropped"] == 0
    return res, table, st


def l2_launches(kt):
    """launches of level 2, whichever record width level 1 wrote (kc_l2_rec6_kernel: the six-byte records of k = 21)"""
    return kt.get("kc_l2_split_kernel", (0, 0.0))[0] + kt.get("kc_l2_rec6_kernel", (0, 0.0))[0]


def assert_same(got, want):
    for g, w, name in zip(got, want, ("keys", "counts", "left", "right")):
        assert g.shape == w.shape, "%s: %s vs %s" % (name, g.shape, w.shape)
        assert (g == w).all(), name


def arrays(reads, quals):
    return O.reads_to_arrays(reads, quals)


# the two insert paths of the library: bucketed (LDS-counted regions, the default) and the global table;
# "small" forces a real two-level partition with several writers on inputs the oracle finishes in seconds
# "compact": one-word k-mers with k <= 23 travel as an invertible mix of the k-mer and the regions hold 32-bit
# records (power-of-two fan-outs that imply enough bits); "wide" switches that off
PATHS = {
    "bucketed": None,
    "bucketed-small": dict(writers=3, p1=4, p2=8, slots=512),
    "bucketed-odd": dict(writers=5, p1=7, p2=13, slots=256),
    "compact": dict(writers=3, p1=256, p2=256, slots=512),
    "compact-short": dict(writers=5, p1=1024, p2=512, slots=256),  # the mix fits 32 bits below the bucket up to k = 21: kc_compact.hpp
    "wide": dict(mode=2, writers=3, p1=256, p2=256, slots=512),
    "table": dict(mode=1),
}


@pytest.mark.parametrize("path", list(PATHS))
@pytest.mark.parametrize("k", [13, 17, 21, 23, 29, 30, 31, 33, 51, 55, 62, 63, 64, 77, 95, 99, 125])  # 30/31 mod 32: extension word
def test_random_reads_match_oracle(k, path):
    rng = np.random.default_rng(100 + k)
    reads, quals = random_reads(rng, 1500, min_len=max(3, k - 5), max_len=k + 130, genome_len=3000)
    b, q, offs = arrays(reads, quals)
    want, wtable, wst = oracle_run(b, q, offs, k)
    with pkg.KmerCounter(k, tuning=PATHS[path]) as kc:
        kc.submit_reads(b, q, offs)
        kc.flush()
        gtable = kc.dump_table()
        got = kc.sorted_results()
        st = kc.stats()
    assert_same(got, want)
    assert (gtable[0] == wtable[0]).all() and (gtable[1] == wtable[1]).all() and (gtable[2] == wtable[2]).all()
    assert st["raw_kmers"] == wst["raw_kmers"]
    assert st["kmers_inserted"] == wst["kmers_inserted"]
    assert st["num_unique"] == wst["unique"]
    assert st["num_purged"] == wst["purged"]
    assert st["total_kmers"] == wst["total_kmers"] and st["sum_counts"] == wst["sum_counts"]
    assert st["num_dropped"] == 0
    assert len(got[1]) > 100


def test_synthetic_arctic_shaped_reads_match_oracle():
    p = pkg.synth_params(num_genomes=8, min_genome_len=30000, max_genome_len=60000, n_rate=0.001)
    b, q, offs = pkg.synth_reads_host(30000, params=p)
    for k in (21, 51):
        want, _, wst = oracle_run(b, q, offs, k, nranks=4, nthreads=4)
        got, st = pkg.analyze_kmers(k, 33, b, q, offs)
        assert_same(got, want)
        assert st["num_unique"] == wst["unique"] and st["sum_counts"] == wst["sum_counts"]


@pytest.mark.parametrize("tuning", [None, PATHS["compact"]], ids=["default", "compact"])
def test_device_resident_input_and_unaligned_pointers(tuning):
    import torch
    k = 21
    rng = np.random.default_rng(7)
    reads, quals = random_reads(rng, 800, min_len=30, max_len=200, genome_len=2500)
    b, q, offs = arrays(reads, quals)
    want, _, _ = oracle_run(b, q, offs, k)
    for shift_b, shift_q in ((0, 0), (1, 1), (7, 7), (15, 15), (3, 9)):
        db = torch.zeros(len(b) + 64, dtype=torch.uint8, device="cuda")
        dq = torch.zeros(len(q) + 64, dtype=torch.uint8, device="cuda")
        db[shift_b:shift_b + len(b)] = torch.from_numpy(b).cuda()
        dq[shift_q:shift_q + len(q)] = torch.from_numpy(q).cuda()
        doff = torch.from_numpy(offs.astype(np.int64)).cuda()
        torch.cuda.synchronize()
        with pkg.KmerCounter(k, tuning=tuning) as kc:
            kc.submit_reads(db[shift_b:], dq[shift_q:], doff, nreads=len(reads))
            assert_same(kc.sorted_results(), want)


@pytest.mark.parametrize("tuning", [None, PATHS["compact"]], ids=["default", "compact"])
def test_seq_block_format_matches_oracle(tuning):
    # ParseAndPackGPUDriver::process_seq_block's input: case-masked reads joined by '_'
    k = 21
    rng = np.random.default_rng(8)
    reads, quals = random_reads(rng, 600, min_len=15, max_len=160, genome_len=2000)
    b, q, offs = arrays(reads, quals)
    want, _, wst = oracle_run(b, q, offs, k)
    masked = []
    for r, ql in zip(reads, quals):
        masked.append("".join(c.lower() if ord(x) < 33 + 20 else c for c, x in zip(r, ql)))
    for block in ("_".join(masked), "_".join(masked) + "_", "_" + "__".join(masked)):
        with pkg.KmerCounter(k, tuning=tuning) as kc:
            kc.submit_seq_block(block.encode())
            got = kc.sorted_results()
            st = kc.stats()
        assert_same(got, want)
        assert st["raw_kmers"] == wst["raw_kmers"]


def test_many_blocks_equal_one_block():
    k = 33
    rng = np.random.default_rng(9)
    reads, quals = random_reads(rng, 900, min_len=20, max_len=150, genome_len=2500)
    b, q, offs = arrays(reads, quals)
    want, _, _ = oracle_run(b, q, offs, k)
    with pkg.KmerCounter(k) as kc:
        for r0 in range(0, 900, 100):
            bb, qq, oo = arrays(reads[r0:r0 + 100], quals[r0:r0 + 100])
            kc.submit_reads(bb, qq, oo)
        assert_same(kc.sorted_results(), want)


def test_sharded_flow_on_one_gpu():
    """kc_extract_partition -> (exchange) -> kc_insert_records with three shards living on one
    device: the union of the shards' results is the oracle's set and no k-mer has two owners."""
    import torch
    for k in (21, 31, 51):
        nl = pkg.lib().kc_record_longs(k)  # words of a record on the wire (31: one more than the k-mer's)
        rng = np.random.default_rng(10 + k)
        reads, quals = random_reads(rng, 1000, min_len=30, max_len=150, genome_len=2500)
        b, q, offs = arrays(reads, quals)
        want, _, wst = oracle_run(b, q, offs, k)
        R = 3
        # k=21: the shards use the benchmark's fan-outs, i.e. compact records (mixed on arrival)
        shards = [pkg.KmerCounter(k, rank_me=r, rank_n=R, tuning=PATHS["compact"] if k == 21 else None) for r in range(R)]
        seg = int(wst["kmers_inserted"])  # worst case: everything to one shard
        recs = torch.zeros(R * seg * nl, dtype=torch.int64, device="cuda")
        # each "rank" parses a third of the reads and bins for all owners
        for part in range(R):
            sl = slice(part * 334, min(1000, (part + 1) * 334))
            bb, qq, oo = arrays(reads[sl], quals[sl])
            counts = shards[part].extract_partition(bb, qq, oo, recs, seg)
            assert int(counts.sum()) == sum(max(0, len(r) - k - 1) for r in reads[sl])
            for d in range(R):
                shards[d].insert_records(recs[d * seg * nl:], int(counts[d]))
                shards[d].flush()
        parts = [s.sorted_results() for s in shards]
        keys = np.concatenate([p[0] for p in parts])
        order = np.lexsort([keys[:, j] for j in range(keys.shape[1] - 1, -1, -1)])
        got = tuple(np.concatenate([p[i] for p in parts])[order] for i in range(4))
        assert_same(got, want)
        L = pkg.lib()
        for r, p in enumerate(parts):
            for i in range(0, len(p[1]), 37):
                kw = np.ascontiguousarray(p[0][i])
                assert L.kc_owner(kw.ctypes.data, k, R) == r
        # too small a segment is reported, not silently truncated
        with pytest.raises(pkg.KcError) as e:
            shards[0].extract_partition(b, q, offs, recs, 10)
        assert e.value.status == -6
        for s in shards:
            s.close()


def test_local_filter_when_sharded():
    # kc_submit_reads with rank_n > 1 inserts only the k-mers this shard owns
    k = 21
    rng = np.random.default_rng(12)
    reads, quals = random_reads(rng, 500, min_len=30, max_len=150, genome_len=2000)
    b, q, offs = arrays(reads, quals)
    want, _, _ = oracle_run(b, q, offs, k)
    parts = []
    for r in range(2):
        with pkg.KmerCounter(k, rank_me=r, rank_n=2) as kc:
            kc.submit_reads(b, q, offs)
            parts.append(kc.sorted_results())
    keys = np.concatenate([p[0] for p in parts])
    order = np.lexsort([keys[:, 0]])
    got = tuple(np.concatenate([p[i] for p in parts])[order] for i in range(4))
    assert_same(got, want)
    assert len(parts[0][1]) > 0 and len(parts[1][1]) > 0


def test_edge_cases():
    k = 21
    with pkg.KmerCounter(k) as kc:  # nothing submitted
        assert len(kc.results()[1]) == 0
    with pkg.KmerCounter(k) as kc:  # empty and too-short reads only
        b, q, offs = arrays(["", "ACGT", "A" * 20, "ACGTACGTACGTACGTACGTA", "ACGTACGTACGTACGTACGTAC"], None)
        kc.submit_reads(b, q, offs)
        st = kc.stats()
        assert len(kc.results()[1]) == 0
        assert st["kmers_inserted"] == 0 and st["raw_kmers"] == 1 + 2
    with pkg.KmerCounter(k) as kc:  # a byte outside ACGTN is an error (reference: DIE)
        b, q, offs = arrays(["ACGTACGTACGTAC_TACGTACGTACGT"], None)
        with pytest.raises(pkg.KcError) as e:
            kc.submit_reads(b, q, offs)
            kc.flush()
        assert e.value.status == -7


def test_hand_cases_small_k():
    # the hand-derived cases of tests/test_oracle_end_to_end.py, through the GPU
    cases = [
        (["ACGTACGGA"] * 2, None, 5, [("CCGTA", 2, "T", "C")]),
        (["AACCGTAG", "CTACGGTT"], None, 5, [("ACCGT", 2, "A", "A"), ("CCGTA", 2, "A", "G")]),
        (["AACCGTAG"] * 2 + ["AACCGTAC"] * 2, None, 5, [("ACCGT", 4, "A", "A")]),
        (["AACCGTAG"] * 2, ["IIIIIIII", "#IIIIIII"], 5, [("CCGTA", 2, "A", "G")]),
        (["AACCNTAG", "AACCGTAG"], None, 5, [("ACCGT", 2, "A", "A"), ("CCGTA", 2, "A", "G")]),
        (["TACGTC"] * 2, None, 4, [("ACGT", 2, "T", "C")]),
    ]
    for reads, quals, k, want in cases:
        b, q, offs = arrays(reads, quals)
        with pkg.KmerCounter(k) as kc:
            kc.submit_reads(b, q, offs)
            lines = kc.dump_lines()
        assert lines == ["%s %d %s %s" % w for w in want], (reads, lines)


def test_count_and_ext_saturation():
    k = 5
    unit = "AACCGTAG"
    n = 70000
    b, q, offs = arrays([unit] * n, None)
    with pkg.KmerCounter(k) as kc:
        kc.submit_reads(b, q, offs)
        keys, counts, exts = kc.dump_table()
        lines = kc.dump_lines()
    assert int(counts.max()) == 65535 and int(exts.max()) == 65535
    assert "ACCGT 65535 A A" in lines


def test_table_grows_instead_of_dropping():
    k = 21
    rng = np.random.default_rng(13)
    # ~400k distinct k-mers into a table created for 1k
    genome = "".join(rng.choice(list("ACGT"), size=200000))
    reads = [genome[i:i + 150] for i in range(0, 200000 - 150, 60)] * 2
    b, q, offs = arrays(reads, None)
    want, _, wst = oracle_run(b, q, offs, k, nranks=2)
    with pkg.KmerCounter(k, max_elems=1000, tuning=dict(mode=1)) as kc:
        cap0 = kc.stats()["capacity"]
        kc.submit_reads(b, q, offs)
        got = kc.sorted_results()
        st = kc.stats()
    assert st["capacity"] > cap0 and st["num_dropped"] == 0
    assert st["num_unique"] == wst["unique"]
    assert_same(got, want)


def test_reset_reuses_the_arena_across_k():
    rng = np.random.default_rng(14)
    reads, quals = random_reads(rng, 600, min_len=40, max_len=180, genome_len=2500)
    b, q, offs = arrays(reads, quals)
    for tuning in (None, dict(mode=1)):
        with pkg.KmerCounter(21, max_elems=200000, max_kmers_buffered=1 << 20, tuning=tuning) as kc:
            sizes = {}
            for k in (21, 33, 55, 77, 21, 33):
                kc.reset(k)
                kc.submit_reads(b, q, offs)
                want, _, _ = oracle_run(b, q, offs, k)
                assert_same(kc.sorted_results(), want)
                # coming back to a record width seen before does not allocate more than the sweep already holds
                sizes.setdefault(kc.nl, kc.stats()["table_bytes"])
                assert kc.stats()["table_bytes"] <= max(sizes.values())


def test_dmin_thres_is_honoured():
    k = 21
    rng = np.random.default_rng(15)
    reads, quals = random_reads(rng, 700, min_len=30, max_len=150, genome_len=1500)
    b, q, offs = arrays(reads, quals)
    for dmin in (1, 2, 3, 5):
        want, _, _ = oracle_run(b, q, offs, k, dmin_thres=dmin)
        with pkg.KmerCounter(k, dmin_thres=dmin) as kc:
            kc.submit_reads(b, q, offs)
            assert_same(kc.sorted_results(), want)


def test_device_and_host_generators_agree():
    import torch
    p = pkg.synth_params(num_genomes=5, min_genome_len=3000, max_genome_len=9000, read_len=150, n_rate=0.002)
    n = 5000
    b, q, offs = pkg.synth_reads_host(n, first_read=123, params=p)
    db = torch.empty(n * 150, dtype=torch.uint8, device="cuda")
    dq = torch.empty(n * 150, dtype=torch.uint8, device="cuda")
    do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    with pkg.KmerCounter(21) as kc:
        kc.synth_reads_device(db, dq, do, n, first_read=123, params=p)
    assert (db.cpu().numpy() == b).all() and (dq.cpu().numpy() == q).all()
    assert (do.cpu().numpy().astype(np.uint64) == offs).all()


# ---- bucketed path: the branches that only unusual data reaches -------------------------------------
def _reads_for_overflow(seed=21, n=1200):
    rng = np.random.default_rng(seed)
    return random_reads(rng, n, min_len=40, max_len=160, genome_len=4000)


@pytest.mark.parametrize("k", [13, 21, 51])  # k=13: compact records with these small power-of-two fan-outs
@pytest.mark.parametrize("tuning", [
    dict(writers=2, p1=2, p2=4, slots=4096, chunk1=16, chain1_max=8, ovf_capacity=1 << 20),   # level-1 chains overflow
    dict(writers=2, p1=2, p2=4, slots=4096, chunk1=16, arena1=20, ovf_capacity=1 << 20),      # a writer's arena runs out
    dict(writers=2, p1=2, p2=4, slots=4096, chunk2=16, chain2_max=10, ovf_capacity=1 << 20),  # region chains overflow
    dict(writers=2, p1=2, p2=4, slots=64),                                                    # more distinct k-mers than LDS slots
    dict(writers=3, p1=4, p2=8, slots=4096, chunk1=16, chunk2=16),                            # many chunks per chain, nothing overflows
    dict(writers=4, p1=8, p2=8, slots=128, chunk1=16, chain1_max=6, chunk2=16, chain2_max=12, ovf_capacity=1 << 20),  # all at once
], ids=["chain1-overflow", "arena1-exhausted", "chain2-overflow", "lds-full", "many-chunks", "everything"])
def test_overflow_paths_are_exact(k, tuning):
    reads, quals = _reads_for_overflow(21 + k)
    b, q, offs = arrays(reads, quals)
    want, wtable, wst = oracle_run(b, q, offs, k)
    with pkg.KmerCounter(k, tuning=tuning) as kc:
        kc.submit_reads(b, q, offs)
        gtable = kc.dump_table()
        got = kc.sorted_results()
        st = kc.stats()
    assert_same(got, want)
    assert (gtable[0] == wtable[0]).all() and (gtable[1] == wtable[1]).all() and (gtable[2] == wtable[2]).all()
    assert st["num_unique"] == wst["unique"] and st["num_purged"] == wst["purged"] and st["sum_counts"] == wst["sum_counts"]


@pytest.mark.parametrize("k", [21, 51])
def test_second_overflow_list_grows_when_level_2_asks(k):
    """The list of region overflows starts small (an eighth of the first list by default) and level 2 is run again with
    the size its counter asked for: regions of at most 160 chained records, 8 of them, 60-100 k occurrences, a list of
    20000 records (more than one tile of positions, so that level 1 may launch)."""
    reads, quals = _reads_for_overflow(77 + k)
    b, q, offs = arrays(reads, quals)
    want, _, wst = oracle_run(b, q, offs, k)
    assert wst["kmers_inserted"] > 40000  # far more than the list and the chains hold
    with pkg.KmerCounter(k, tuning=dict(writers=2, p1=2, p2=4, slots=4096, chunk2=16, chain2_max=10, ovf_capacity=20000)) as kc:
        kc.submit_reads(b, q, offs)
        got = kc.sorted_results()
        st = kc.stats()
    assert_same(got, want)
    assert st["num_unique"] == wst["unique"] and st["sum_counts"] == wst["sum_counts"]


@pytest.mark.parametrize("k", [21, 51])
def test_chain_positions_across_chunks_and_launches(k):
    """The arithmetic behind the one GPU fault on record (round 2, gpurun_out/r2_b2.err: an uncommitted build of the
    split kernels added a run's `index minus staging position`, which is kept modulo 2^32, to the staging position in
    64 bits -- 16 GiB past the arena whenever a run starts at a smaller chain position than its place in the staging,
    i.e. in every first round).  Chains that continue a partly filled last chunk from an earlier
    launch and run on into new chunks, at both levels, with 16-record chunks so that every chain crosses many of them,
    several launches with different run lengths; compact and wide records."""
    rng = np.random.default_rng(400 + k)
    reads, quals = random_reads(rng, 3000, min_len=k + 2, max_len=k + 120, genome_len=4000)
    b, q, offs = arrays(reads, quals)
    want, wtable, wst = oracle_run(b, q, offs, k)
    cuts = [0, 7, 300, 301, 1500, 2999, 3000]  # launches of 7, 293, 1, 1199, 1499 and 1 reads
    for tuning in (dict(writers=2, p1=4, p2=4, slots=4096, chunk1=16, chunk2=16), dict(writers=3, p1=8, p2=16, slots=2048, chunk1=16, chunk2=32)):
        with pkg.KmerCounter(k, tuning=tuning) as kc:
            for a, z in zip(cuts[:-1], cuts[1:]):
                bb, qq, oo = arrays(reads[a:z], quals[a:z])
                kc.submit_reads(bb, qq, oo)
            gtable = kc.dump_table()
            got = kc.sorted_results()
            st = kc.stats()
        assert_same(got, want)
        assert (gtable[0] == wtable[0]).all() and (gtable[1] == wtable[1]).all() and (gtable[2] == wtable[2]).all()
        assert st["num_unique"] == wst["unique"] and st["sum_counts"] == wst["sum_counts"]


def test_result_arrays_grow_when_the_estimate_was_too_small():
    # max_elems far below the truth: the first vote/purge pass finds the result arrays too small (its blocks of
    # output positions run past them) and is run again with room for what it counted
    k = 21
    rng = np.random.default_rng(77)
    reads, quals = random_reads(rng, 24000, min_len=100, max_len=160, genome_len=150000, err=0.005)
    b, q, offs = arrays(reads, quals)
    want, _, wst = oracle_run(b, q, offs, k)
    assert len(want[1]) > 110000  # more survivors than the first allocation holds
    with pkg.KmerCounter(k, max_elems=2000, max_kmers_buffered=1 << 23, tuning=dict(p1=64, p2=64, slots=2048)) as kc:
        kc.submit_reads(b, q, offs)
        got = kc.sorted_results()
        st = kc.stats()
    assert_same(got, want)
    assert st["num_unique"] == wst["unique"] and st["num_dropped"] == 0


def test_exhausted_overflow_list_is_an_error_not_a_loss():
    reads, quals = _reads_for_overflow()
    b, q, offs = arrays(reads, quals)
    with pkg.KmerCounter(21, tuning=dict(writers=2, p1=2, p2=4, chunk1=16, chain1_max=4, ovf_capacity=50)) as kc:
        with pytest.raises(pkg.KcError) as e:
            kc.submit_reads(b, q, offs)
            kc.results()
        assert e.value.status == -6


def test_full_buffer_falls_back_to_the_table_path():
    k = 21
    reads, quals = _reads_for_overflow(5, n=2000)
    b, q, offs = arrays(reads, quals)
    want, _, wst = oracle_run(b, q, offs, k)
    with pkg.KmerCounter(k, max_kmers_buffered=50000) as kc:  # far fewer than the ~170k occurrences
        for r0 in range(0, 2000, 250):
            bb, qq, oo = arrays(reads[r0:r0 + 250], quals[r0:r0 + 250])
            kc.submit_reads(bb, qq, oo)
        got = kc.sorted_results()
        st = kc.stats()
    assert_same(got, want)
    assert st["num_unique"] == wst["unique"] and st["kmers_inserted"] == wst["kmers_inserted"]


@pytest.mark.parametrize("k,tuning", [(21, None), (21, dict(writers=3, p1=256, p2=256, slots=512)), (33, dict(ovf_capacity=1 << 16)),
                                      (77, dict(writers=3, p1=4, p2=8, slots=512, ovf_capacity=1 << 16))],  # (the lists of so small a buffer would not hold one tile)
                         ids=["k21", "k21-compact", "k33", "k77-small"])
def test_a_full_buffer_is_counted_and_merged_not_abandoned(k, tuning):
    """More reads than max_kmers_buffered holds: every time the buffer is full its k-mers are counted, merged into the
    global table (one table operation per distinct k-mer) and the buffer starts again empty -- the context stays on the
    fast path (the reference streams insert blocks into its one table indefinitely, gpu_hash_table.cpp:681-695).
    Several passes, a pre-purge dump in the middle of it all, the result bit-exact."""
    reads, quals = _reads_for_overflow(50 + k, n=2400)
    b, q, offs = arrays(reads, quals)
    want, wtable, wst = oracle_run(b, q, offs, k)
    per_block = sum(max(0, len(r) - k - 1) for r in reads[:300])
    with pkg.KmerCounter(k, max_kmers_buffered=int(per_block * 2.5), tuning=tuning, time_kernels=True) as kc:  # two blocks fit, the third does not
        for r0 in range(0, 2400, 300):
            bb, qq, oo = arrays(reads[r0:r0 + 300], quals[r0:r0 + 300])
            kc.submit_reads(bb, qq, oo)
        gtable = kc.dump_table()
        got = kc.sorted_results()
        st = kc.stats()
        kt = kc.kernel_times()
    assert_same(got, want)
    assert (gtable[0] == wtable[0]).all() and (gtable[1] == wtable[1]).all() and (gtable[2] == wtable[2]).all()
    assert st["num_unique"] == wst["unique"] and st["num_purged"] == wst["purged"] and st["sum_counts"] == wst["sum_counts"]
    assert st["kmers_inserted"] == wst["kmers_inserted"]
    # the reads went through level 1 every time (no launch of the global-table extraction kernel), in several passes
    assert "kc_extract_kernel<insert>" not in kt and l2_launches(kt) >= 3


@pytest.mark.parametrize("light", [True, False], ids=["light", "merged"])
@pytest.mark.parametrize("path", ["compact", "compact-short"])
@pytest.mark.parametrize("blocks_fit", [2.5, 1.2])
def test_compact_records_leave_level_1_when_the_buffer_is_full(path, light, blocks_fit, monkeypatch):
    """Compact records and a buffer smaller than the input: what level 1 holds goes through level 2 (an instalment) and
    level 1 starts again empty -- level 2 keeps its 4-byte records until the regions are counted once, at the end, growing
    its arena as the input grows (bk_light_spill, kc_l2_grow_kernel: with room for 1.2 blocks of eight the arena moves
    several times).  Nothing is merged into the global table; KC_LIGHT_SPILL=0 takes the other way (the counted buffer
    merged into the table, what longer k-mers do).  Bit-exact either way, the pre-purge dump too."""
    if not light:
        monkeypatch.setenv("KC_LIGHT_SPILL", "0")
    k = 21
    reads, quals = _reads_for_overflow(150, n=2400)
    b, q, offs = arrays(reads, quals)
    want, wtable, wst = oracle_run(b, q, offs, k)
    per_block = sum(max(0, len(r) - k - 1) for r in reads[:300])
    with pkg.KmerCounter(k, max_kmers_buffered=int(per_block * blocks_fit), tuning=PATHS[path], time_kernels=True) as kc:
        for r0 in range(0, 2400, 300):
            bb, qq, oo = arrays(reads[r0:r0 + 300], quals[r0:r0 + 300])
            kc.submit_reads(bb, qq, oo)
        gtable = kc.dump_table()
        got = kc.sorted_results()
        st = kc.stats()
        kt = kc.kernel_times()
    assert_same(got, want)
    assert (gtable[0] == wtable[0]).all() and (gtable[1] == wtable[1]).all() and (gtable[2] == wtable[2]).all()
    assert st["num_unique"] == wst["unique"] and st["num_purged"] == wst["purged"] and st["sum_counts"] == wst["sum_counts"]
    assert st["kmers_inserted"] == wst["kmers_inserted"]
    assert "kc_extract_kernel<insert>" not in kt and l2_launches(kt) >= 3
    # the light way merges no counted buffer into the global table (kc_merge_entries_kernel is timed as
    # kc_insert_records_kernel; so is the one move of the overflow records at the end, which these reads provoke)
    merges = kt.get("kc_insert_records_kernel", (0, 0.0))[0]
    assert merges <= 1 if light else merges >= 2, kt


def test_records_path_spills_a_full_buffer_too():
    k, R = 21, 2
    rng = np.random.default_rng(23)
    reads, quals = random_reads(rng, 1600, min_len=30, max_len=150, genome_len=3000)
    b, q, offs = arrays(reads, quals)
    want, _, wst = oracle_run(b, q, offs, k)
    import torch
    total = int(wst["kmers_inserted"])
    shards = [pkg.KmerCounter(k, rank_me=r, rank_n=R, max_kmers_buffered=total // 6, time_kernels=True) for r in range(R)]
    recs = torch.zeros(R * total, dtype=torch.int64, device="cuda")
    for r0 in range(0, 1600, 200):  # eight blocks of about total / 8 records, half of each per shard
        bb, qq, oo = arrays(reads[r0:r0 + 200], quals[r0:r0 + 200])
        counts = shards[0].extract_partition(bb, qq, oo, recs, total)
        for d in range(R):
            shards[d].insert_records(recs[d * total:], int(counts[d]))
    parts = [s.sorted_results() for s in shards]
    keys = np.concatenate([p[0] for p in parts])
    order = np.lexsort([keys[:, 0]])
    assert_same(tuple(np.concatenate([p[i] for p in parts])[order] for i in range(4)), want)
    for s in shards:
        kt = s.kernel_times()
        assert l2_launches(kt) >= 2  # more than one pass
        s.close()


def test_heavy_hitter_kmers():
    # one k-mer in almost every read: a region far above its mean size, counters far above the rest
    k = 21
    rng = np.random.default_rng(33)
    rep = "ACGTTGCATGCATGCCGATTACGG"
    reads = []
    for _ in range(3000):
        tail = "".join(rng.choice(list("ACGT"), size=40))
        reads.append(tail[:20] + rep + tail[20:])
    b, q, offs = arrays(reads, None)
    want, _, _ = oracle_run(b, q, offs, k)
    for tuning in (None, dict(writers=4, p1=4, p2=4, slots=2048)):
        with pkg.KmerCounter(k, tuning=tuning) as kc:
            kc.submit_reads(b, q, offs)
            assert_same(kc.sorted_results(), want)


def pack_reads(reads, quals, qual_offset=33):
    """The reference's read cache bytes (PackedRead, src/packed_reads.cpp:99-126): base | min(q - offset, 31) << 3."""
    code = {"A": 0, "C": 1, "G": 2, "T": 3, "N": 4}
    out = bytearray()
    for r, q in zip(reads, quals):
        out.extend(code[c] | (min(ord(x) - qual_offset, 31) << 3) for c, x in zip(r, q))
    return np.frombuffer(bytes(out), dtype=np.uint8).copy()


@pytest.mark.parametrize("k", [21, 51])
@pytest.mark.parametrize("tuning", [None, dict(mode=1)], ids=["bucketed", "table"])
def test_packed_reads_ingest_matches_oracle(k, tuning):
    import torch
    rng = np.random.default_rng(40 + k)
    reads, quals = random_reads(rng, 1200, min_len=k - 2, max_len=k + 120, genome_len=2500)
    b, q, offs = arrays(reads, quals)
    want, _, wst = oracle_run(b, q, offs, k)
    packed = pack_reads(reads, quals)
    with pkg.KmerCounter(k, tuning=tuning) as kc:  # host-resident
        kc.submit_packed_reads(packed, offs)
        got = kc.sorted_results()
        st = kc.stats()
    assert_same(got, want)
    assert st["raw_kmers"] == wst["raw_kmers"] and st["num_unique"] == wst["unique"]
    with pkg.KmerCounter(k, tuning=tuning) as kc:  # device-resident, odd alignment
        dp = torch.zeros(len(packed) + 32, dtype=torch.uint8, device="cuda")
        dp[5:5 + len(packed)] = torch.from_numpy(packed).cuda()
        kc.submit_packed_reads(dp[5:], torch.from_numpy(offs.astype(np.int64)).cuda(), nreads=len(reads))
        assert_same(kc.sorted_results(), want)
    with pkg.KmerCounter(k, tuning=tuning) as kc:  # a base code above 4 is an error
        bad = packed.copy()
        bad[100] = (bad[100] & 0xF8) | 6
        with pytest.raises(pkg.KcError) as e:
            kc.submit_packed_reads(bad, offs)
            kc.results()
        assert e.value.status == -7


@pytest.mark.parametrize("k", [21, 51, 77])
def test_reference_owner_mode_bins_like_get_kmer_target_rank(k):
    """KC_FLAG_REFERENCE_OWNER: every record lands in the segment of the rank the reference's own
    KmerDHT::get_kmer_target_rank would send it to (oracle restatement, pinned by the SURVEY known answers), and the
    sharded flow still reproduces the oracle's result."""
    import torch
    nl = pkg.lib().kc_num_longs(k)
    m = O.lib().orc_minimizer_len(k)
    rng = np.random.default_rng(60 + k)
    reads, quals = random_reads(rng, 600, min_len=k, max_len=k + 100, genome_len=2000)
    b, q, offs = arrays(reads, quals)
    want, _, wst = oracle_run(b, q, offs, k)
    R = 5
    shards = [pkg.KmerCounter(k, rank_me=r, rank_n=R, reference_owner=True) for r in range(R)]
    seg = int(wst["kmers_inserted"])
    recs = torch.zeros(R * seg * nl, dtype=torch.int64, device="cuda")
    counts = shards[0].extract_partition(b, q, offs, recs, seg)
    assert int(counts.sum()) == wst["kmers_inserted"] and (counts > 0).all()
    host = recs.cpu().numpy().view(np.uint64).reshape(R, seg, nl)
    for d in range(R):
        seg_recs = host[d, :int(counts[d])].copy()
        seg_recs[:, nl - 1] &= ~np.uint64(0x3F)
        for row in seg_recs[::53]:
            assert O.target_rank(np.ascontiguousarray(row), k, m, R) == d
        shards[d].insert_records(recs[d * seg * nl:], int(counts[d]))
    parts = [s.sorted_results() for s in shards]
    keys = np.concatenate([p[0] for p in parts])
    order = np.lexsort([keys[:, j] for j in range(nl - 1, -1, -1)])
    got = tuple(np.concatenate([p[i] for p in parts])[order] for i in range(4))
    assert_same(got, want)
    for s in shards:
        s.close()


@pytest.mark.parametrize("k", [21, 31, 51, 63])
def test_lookup_over_results(k):
    """kc_lookup = KmerDHT::get_kmer_count in bulk: every result is found with its count and extensions from either
    strand; purged and never-seen k-mers come back with count 0."""
    rng = np.random.default_rng(70 + k)
    reads, quals = random_reads(rng, 800, min_len=k + 2, max_len=k + 100, genome_len=2500)
    b, q, offs = arrays(reads, quals)
    want, wtable, _ = oracle_run(b, q, offs, k)
    with pkg.KmerCounter(k) as kc:
        kc.submit_reads(b, q, offs)
        keys, counts, left, right = kc.sorted_results()
        c, l, r = kc.lookup(keys)
        assert (c == counts).all() and (l == left).all() and (r == right).all()
        rcq = np.stack([O.revcomp(keys[i], k) for i in range(0, len(keys), 7)])  # the other strand finds the same entry
        c2, l2, r2 = kc.lookup(rcq)
        assert (c2 == counts[::7]).all() and (l2 == left[::7]).all()
        # k-mers that were counted but purged, and random ones: absent
        kept = {tuple(int(x) for x in row) for row in keys}
        purged = np.stack([row for row in wtable[0] if tuple(int(x) for x in row) not in kept][:500])
        c3, _, _ = kc.lookup(purged)
        assert (c3 == 0).all()
        rnd = np.stack([O.pack_kmer("".join(rng.choice(list("ACGT"), size=k))) for _ in range(200)])
        c4, _, _ = kc.lookup(rnd)
        assert (c4 == 0).sum() >= 199
        c5, _, _ = kc.lookup(np.zeros((0, kc.nl), dtype=np.uint64))
        assert len(c5) == 0


def test_heavy_regions_through_a_small_buffer_grow_the_overflow_list():
    """k = 11 has few distinct k-mers, so every region's chain is long and the regions that outgrow theirs fill the
    region overflow list INSIDE an instalment behind which level 1 is emptied (bk_light_spill) -- level 2's state is put
    back, the list grown, the instalment run again; the same for the last pass (scripts/stress_parity.py found the case)"""
    k = 11
    rng = np.random.default_rng(384)
    reads, quals = random_reads(rng, 5200, min_len=20, max_len=180, genome_len=10800, err=0.005, n_rate=0.0)
    b, q, offs = arrays(reads, quals)
    want, wtable, wst = oracle_run(b, q, offs, k)
    occ = int(wst["kmers_inserted"])
    for cap in (40000, occ // 5):
        with pkg.KmerCounter(k, max_kmers_buffered=cap) as kc:
            for r0 in range(0, len(reads), 400):
                bb, qq, oo = arrays(reads[r0:r0 + 400], quals[r0:r0 + 400])
                kc.submit_reads(bb, qq, oo)
            gtable = kc.dump_table()
            got = kc.sorted_results()
            st = kc.stats()
        assert_same(got, want)
        assert (gtable[0] == wtable[0]).all() and (gtable[1] == wtable[1]).all() and (gtable[2] == wtable[2]).all()
        assert st["kmers_inserted"] == occ
