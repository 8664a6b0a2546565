"""End-to-end oracle (C) against the independent Python spec model and against
hand-derived cases for S1-S9."""
import numpy as np
import pytest

import spec_model as M
from helpers import random_reads, results_to_tuples
from oracle import cpu_oracle as O


def oracle_results(reads, quals, k, **kw):
    (keys, counts, left, right), st = O.count_reads(reads, quals, k=k, **kw)
    res = [(O.kmer_to_string(keys[i], k), int(counts[i]), chr(left[i]), chr(right[i])) for i in range(len(counts))]
    return res, st


def model_sorted_by_packed(results):
    # the model sorts by string; ACGT string order == packed order, so they agree
    return sorted(results)


@pytest.mark.parametrize("k", [21, 31, 33, 51, 77])
@pytest.mark.parametrize("nranks,nthreads", [(1, 1), (3, 2), (8, 4)])
def test_random_reads_match_model(k, nranks, nthreads):
    rng = np.random.default_rng(1000 + k + nranks)
    reads, quals = random_reads(rng, 300, min_len=k - 3, max_len=k + 90, genome_len=700)
    got, st = oracle_results(reads, quals, k, nranks=nranks, nthreads=nthreads)
    want, table = M.count_kmers(reads, quals, k)
    assert st["dropped"] == 0
    assert got == model_sorted_by_packed(want)
    assert st["unique"] == len(table)
    assert st["raw_kmers"] == sum(len(r) - k + 1 for r in reads if len(r) >= k)
    assert st["kmers_inserted"] == sum(max(0, len(r) - k - 1) for r in reads)
    assert st["sum_counts"] == sum(c for _, c, _, _ in want)
    assert len(got) > 10  # the case is not vacuous


def test_full_table_matches_model():
    k = 21
    rng = np.random.default_rng(5)
    reads, quals = random_reads(rng, 200, min_len=30, max_len=100, genome_len=400)
    o = O.Oracle(k, nranks=4, nthreads=2)
    o.add_reads(*O.reads_to_arrays(reads, quals))
    keys, counts, exts = o.dump_table()
    _, table = M.count_kmers(reads, quals, k)
    got = {O.kmer_to_string(keys[i], k): [int(counts[i]), [int(x) for x in exts[i][:4]], [int(x) for x in exts[i][4:]]]
           for i in range(len(counts))}
    assert got == table


def test_result_independent_of_rank_count_and_order():
    k = 21
    rng = np.random.default_rng(6)
    reads, quals = random_reads(rng, 400, min_len=25, max_len=150, genome_len=900)
    base, _ = oracle_results(reads, quals, k, nranks=1)
    for nranks in (2, 5, 16):
        assert oracle_results(reads, quals, k, nranks=nranks, nthreads=3)[0] == base
    perm = rng.permutation(len(reads))
    assert oracle_results([reads[i] for i in perm], [quals[i] for i in perm], k, nranks=3)[0] == base


# ---- hand-derived cases (k=5 keeps them checkable by eye) -------------------

def test_clean_repeat_and_self_overlap():
    # ACGTACGGA, k=5, k-mers with both neighbours are i=1..3:
    #   i=1 CGTAC  (l A, r G)             canonical as is (rc GTACG is larger)
    #   i=2 GTACG  (l C, r G) -> rc CGTAC (l comp(G)=C, r comp(C)=G)
    #   i=3 TACGG  (l G, r A) -> rc CCGTA (l comp(A)=T, r comp(G)=C)
    # two copies: CGTAC count 4 with left votes A:2 C:2 -> fork -> purged;
    #             CCGTA count 2, left T:2, right C:2 -> kept
    read = "ACGTACGGA"
    res, st = oracle_results([read, read], None, 5)
    assert res == [("CCGTA", 2, "T", "C")]
    assert st["kmers_inserted"] == 6 and st["unique"] == 2 and st["purged"] == 1


def test_single_occurrence_is_purged():
    res, st = oracle_results(["AACCGTAG"], None, 5)  # ACCGT and CCGTA, once each
    assert res == [] and st["unique"] == 2 and st["purged"] == 2


def test_reverse_strand_duplicates_swap_and_complement_exts():
    fwd = "AACCGTAG"          # k=5, i=1..2: ACCGT (l A r A), CCGTA (l A r G)
    rev = M.revcomp_str(fwd)  # CTACGGTT: TACGG == rc(CCGTA), ACGGT == rc(ACCGT)
    res, _ = oracle_results([fwd, rev], None, 5)
    assert ("ACCGT", 2, "A", "A") in res
    assert ("CCGTA", 2, "A", "G") in res
    assert len(res) == 2


def test_fork_is_purged():
    a = "AACCGTAG"
    b = "AACCGTAC"  # CCGTA now followed by C twice and G twice -> F
    res, _ = oracle_results([a, a, b, b], None, 5)
    assert [r[0] for r in res] == ["ACCGT"]
    assert res[0] == ("ACCGT", 4, "A", "A")


def test_low_quality_neighbour_is_not_an_extension():
    read = "AACCGTAG"
    q_ok = "IIIIIIII"
    q_lo = "#IIIIIII"  # the left neighbour of ACCGT is low quality in one copy
    res, st = oracle_results([read, read], [q_ok, q_lo], 5)
    # ACCGT: left A seen once (< dmin 2) -> X -> purged; CCGTA still fine
    assert res == [("CCGTA", 2, "A", "G")]
    # the k-mer itself is counted regardless of its own bases' quality
    res2, _ = oracle_results([read, read, read], [q_ok, q_ok, "I###III#"], 5)
    assert ("ACCGT", 3, "A", "A") in res2


def test_n_inside_kmer_counts_as_g_and_n_neighbour_is_ignored():
    res, _ = oracle_results(["AACCNTAG", "AACCGTAG"], None, 5)
    assert ("ACCGT", 2, "A", "A") in res
    # N as the right neighbour of ACCGT: ignored, so right has only one vote -> X
    res, _ = oracle_results(["AACCGTNG", "AACCGTAG"], None, 5)
    assert all(r[0] != "ACCGT" for r in res)


def test_short_reads_contribute_nothing():
    k = 5
    for ln in (0, 1, 4, 5, 6):
        res, st = oracle_results(["ACGTAC"[:ln]] * 3, None, k)
        assert res == [] and st["kmers_inserted"] == 0
    # raw k-mers still counts len-k+1 for len >= k (kcount.cpp:86)
    _, st = oracle_results(["ACGTAC"], None, k)
    assert st["raw_kmers"] == 2


def test_empty_input():
    res, st = oracle_results([], None, 21)
    assert res == [] and st["reads"] == 0


def test_count_saturates_at_65535():
    k = 5
    unit = "AACCGTAG"
    n = 66000
    res, st = oracle_results([unit] * n, None, k, nthreads=4, nranks=2)
    assert ("ACCGT", 65535, "A", "A") in res
    _, table = M.count_kmers([unit] * 3, ["I" * 8] * 3, k)
    o = O.Oracle(k)
    o.add_reads(*O.reads_to_arrays([unit] * n))
    keys, counts, exts = o.dump_table()
    assert int(counts.max()) == 65535 and int(exts.max()) == 65535


def test_palindrome_is_not_swapped():
    # k even allows kmer == rc: ACGT; strict < keeps the forward exts
    res, _ = oracle_results(["TACGTC", "TACGTC"], None, 4)
    assert ("ACGT", 2, "T", "C") in res


def test_bad_character_is_an_error():
    o = O.Oracle(5)
    with pytest.raises(RuntimeError):
        o.add_reads(*O.reads_to_arrays(["ACGT_ACGTA"]))


def test_receiver_entry_matches_whole_path():
    # feeding the receiver one supermer by hand == sending the read through the sender
    k = 5
    o = O.Oracle(k, nranks=2)
    for _ in range(2):
        o.insert_supermer(1, "aACCGTAG")  # lowercase = low quality
    keys, counts, left, right = o.finalize()
    got = [(O.kmer_to_string(keys[i], k), int(counts[i]), chr(left[i]), chr(right[i])) for i in range(len(counts))]
    assert got == [("CCGTA", 2, "A", "G")]
