"""The contig k-mer pass (kc_begin_ctg_kmers / kc_submit_ctg_block, csrc/kc_ctg.hpp) against the oracle's
statement-for-statement restatement of insert_supermer_from_ctg (kcount_cpu.cpp:357-407): reads first, then contigs in
several orders -- the outcome must not depend on the order, and must be the device's."""
import numpy as np
import pytest

import mhm2_kmer_analysis_v2_amd as pkg
from helpers import random_reads
from oracle import cpu_oracle as O
from test_gpu_parity import arrays, assert_same

pytestmark = pytest.mark.gpu


def make_ctgs(rng, genome, k):
    """Contigs cut from the reads' genome (so that many of their k-mers meet read k-mers: kept ones, singletons, forks),
    some of them overlapping each other with equal and with different depths, some with a changed base (another
    extension for the same k-mer), one with an N, one with a lower-case base, one of depth 1 and one of depth 0."""
    ctgs, depths = [], []
    n = len(genome)
    for i in range(40):
        a = int(rng.integers(0, n - 400))
        ln = int(rng.integers(k + 2, 300))
        s = genome[a:a + ln]
        if rng.random() < 0.5:
            s = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
        ctgs.append(s)
        depths.append(int(rng.integers(2, 60)))
    # the same stretch again at other depths, and once with one base changed
    for i in (0, 3, 7):
        ctgs.append(ctgs[i])
        depths.append(depths[i] + 5)
        ctgs.append(ctgs[i][: len(ctgs[i]) // 2])
        depths.append(max(2, depths[i] - 1))
        c = list(ctgs[i])
        j = len(c) // 3
        c[j] = "ACGT"[("ACGT".index(c[j]) + 1) % 4]
        ctgs.append("".join(c))
        depths.append(9)
    c = list(ctgs[1]); c[len(c) // 2] = "N"; ctgs.append("".join(c)); depths.append(7)
    c = list(ctgs[2]); c[k + 3] = c[k + 3].lower(); ctgs.append("".join(c)); depths.append(11)
    ctgs.append(ctgs[4]); depths.append(1)
    ctgs.append(ctgs[5]); depths.append(0)
    # sequence no read covers
    ctgs.append("".join(rng.choice(list("ACGT"), size=500))); depths.append(13)
    return ctgs, depths


@pytest.mark.parametrize("k", [21, 31, 51, 77])
def test_contig_pass_matches_the_oracle_in_every_order(k):
    rng = np.random.default_rng(900 + k)
    genome = "".join(rng.choice(list("ACGT"), size=3000))
    reads, quals = [], []
    for _ in range(900):
        a = int(rng.integers(0, len(genome) - 160))
        ln = int(rng.integers(k + 2, 150))
        s = list(genome[a:a + ln])
        for j in range(ln):
            if rng.random() < 0.01:
                s[j] = "ACGT"[int(rng.integers(0, 4))]
        reads.append("".join(s))
        quals.append("".join("I" if rng.random() > 0.03 else "#" for _ in range(ln)))
    b, q, offs = arrays(reads, quals)
    ctgs, depths = make_ctgs(rng, genome, k)
    want = None
    for order in range(3):
        perm = rng.permutation(len(ctgs)) if order else np.arange(len(ctgs))
        o = O.Oracle(k, nranks=3, nthreads=1)
        o.add_reads(b, q, offs)
        for i in perm:
            o.add_ctg(ctgs[i], depths[i])
        res = o.finalize()
        o.close()
        if want is None:
            want = res
        else:  # the reference's outcome does not depend on the order of the contigs
            for g, w in zip(res, want):
                assert g.shape == w.shape and (g == w).all()
    o = O.Oracle(k, nranks=3, nthreads=1)
    o.add_reads(b, q, offs)
    plain = o.finalize()
    o.close()
    assert len(want[1]) > len(plain[1]) + 50  # the contigs add k-mers
    for tuning in (None, dict(mode=1)):
        with pkg.KmerCounter(k, tuning=tuning) as kc:
            kc.submit_reads(b, q, offs)
            kc.begin_ctg_kmers(sum(len(c) for c in ctgs))
            half = len(ctgs) // 2
            kc.submit_ctgs(ctgs[:half], depths[:half])
            kc.submit_ctgs(ctgs[half:], depths[half:])
            got = kc.sorted_results()
            st = kc.stats()
            assert_same(got, want)
            assert st["total_kmers"] == len(want[1]) and st["sum_counts"] == int(want[1].astype(np.int64).sum())
            # the results can be looked up afterwards (the index is rebuilt over the merged set)
            cnt, _, _ = kc.lookup(want[0][::37])
            assert (cnt == want[1][::37]).all()


def test_contig_pass_errors():
    k = 21
    with pkg.KmerCounter(k) as kc:
        with pytest.raises(pkg.KcError) as e:  # not begun
            kc.submit_ctgs(["ACGT" * 20], [5])
        assert e.value.status == -8
        kc.begin_ctg_kmers(1000)
        with pytest.raises(pkg.KcError) as e:  # a character the reference DIEs on
            kc.submit_ctgs(["ACGT" * 10 + "X" + "ACGT" * 10], [5])
        assert e.value.status == -7


def test_a_contig_table_that_is_too_small_reports_capacity_and_never_spins():
    """kc_begin_ctg_kmers sized from an estimate that is far too low: a block with more distinct k-mers than the table has
    slots must come back with KC_ERR_CAPACITY (launches are bounded by the table's free room), not hang in a probe loop."""
    k = 21
    rng = np.random.default_rng(77)
    ctg = "".join(rng.choice(list("ACGT"), size=6000))  # ~6000 distinct k-mers
    with pkg.KmerCounter(k) as kc:
        kc.begin_ctg_kmers(10)  # -> 1024 slots
        with pytest.raises(pkg.KcError) as e:
            kc.submit_ctgs([ctg], [5])
        assert e.value.status == -6  # KC_ERR_CAPACITY
        distinct, _ = kc.ctg_stats()
        assert 0 < distinct <= 1024 * 3 // 4
    # a table with room takes the same block in one go, and one that is only just large enough takes it in several launches
    for room in (8000, 3000):
        with pkg.KmerCounter(k) as kc:
            kc.begin_ctg_kmers(room)
            kc.submit_ctgs([ctg], [5])
            distinct, positions = kc.ctg_stats()
            assert distinct == len({min(ctg[i:i + k], ctg[i:i + k][::-1].translate(str.maketrans("ACGT", "TGCA"))) for i in range(1, len(ctg) - k)})
            assert positions == len(ctg) + 1


def test_a_bad_character_is_seen_wherever_it_sits():
    """The reference DIEs on any character outside ACGTN (kcount_cpu.cpp:481-487): also in a contig shorter than k + 2 and
    within k + 1 of a contig's end, where no k-mer window reaches it."""
    k = 21
    for ctgs in (["ACGTX"], ["ACGT" * 20, "ACG", "AC!T" + "A" * 5], ["ACGT" * 20 + "Z"], ["Z" + "ACGT" * 20]):
        with pkg.KmerCounter(k) as kc:
            kc.begin_ctg_kmers(1000)
            with pytest.raises(pkg.KcError) as e:
                kc.submit_ctgs(ctgs, [5] * len(ctgs))
            assert e.value.status == -7  # KC_ERR_BAD_BASE


@pytest.mark.parametrize("mode", ["hash", "reference", "shard-flow"])
@pytest.mark.parametrize("k", [21, 51])
def test_contig_pass_of_several_ranks_keeps_every_kmer_exactly_once(k, mode):
    """rank_n > 1: every rank is given every read and every contig; each keeps its own share (the read path's owner test),
    and the union of the ranks' results is the oracle's single answer -- no k-mer twice, none with a contig's depth where
    another rank's reads kept it."""
    rng = np.random.default_rng(1300 + k)
    genome = "".join(rng.choice(list("ACGT"), size=3000))
    reads, quals = [], []
    for _ in range(700):
        a = int(rng.integers(0, len(genome) - 160))
        ln = int(rng.integers(k + 2, 150))
        reads.append(genome[a:a + ln])
        quals.append("I" * ln)
    b, q, offs = arrays(reads, quals)
    ctgs, depths = make_ctgs(rng, genome, k)
    o = O.Oracle(k, nranks=3, nthreads=1)
    o.add_reads(b, q, offs)
    for c, d in zip(ctgs, depths):
        o.add_ctg(c, d)
    want = o.finalize()
    o.close()
    R = 3
    parts = []
    if mode == "shard-flow":
        # every shard extracts its slice of the reads and ships the buckets it does not own; then every shard is given
        # every contig and keeps the k-mers of its own buckets
        from test_gpu_shard_flow import run_shards
        shards, _, _ = run_shards(reads, quals, k, R, None)
        for s in shards:
            s.begin_ctg_kmers(sum(len(c) for c in ctgs))
            s.submit_ctgs(ctgs, depths)
            parts.append(s.sorted_results())
            s.close()
    else:
        for r in range(R):
            with pkg.KmerCounter(k, rank_me=r, rank_n=R, reference_owner=(mode == "reference")) as kc:
                kc.submit_reads(b, q, offs)
                kc.begin_ctg_kmers(sum(len(c) for c in ctgs))
                kc.submit_ctgs(ctgs, depths)
                parts.append(kc.sorted_results())
    keys = np.concatenate([p[0] for p in parts])
    cnt = np.concatenate([p[1] for p in parts])
    lf = np.concatenate([p[2] for p in parts])
    rt = np.concatenate([p[3] for p in parts])
    order = np.lexsort([keys[:, j] for j in range(keys.shape[1] - 1, -1, -1)])
    assert_same((keys[order], cnt[order], lf[order], rt[order]), want)
