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
