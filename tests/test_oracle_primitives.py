"""Oracle primitives against (a) the known answers SURVEY.md section 8c recorded
from the reference's own src/kmer.cpp + src/hash_funcs.c, and (b) oracle/_ref,
the reference's src/hash_funcs.c compiled unmodified (present only where
/root/reference was available at build time)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import cpu_oracle as O

REF_SO = os.path.join(os.path.dirname(O.__file__), "_ref", "libref_hash_funcs.so")


def test_pack_and_revcomp_k21():
    w = O.pack_kmer("ACGTTGCATGCATGCCGATTA")
    assert int(w[0]) == 0x1BE4E4E58F000000
    assert int(O.revcomp(w, 21)[0]) == 0xC369393906C00000
    assert int(O.pack_kmer("TAATCGGCATGCATGCAACGT")[0]) == 0xC369393906C00000


def test_n_becomes_g_and_hashes_k21():
    # read ...CCGATNACG...: the k-mer over the N packs as if it were G
    kms = O.get_kmers("ACGTTGCATGCATGCCGATNACG", 21)
    assert (kms[0] == O.pack_kmer("ACGTTGCATGCATGCCGATGA")).all()
    assert O.kmer_hash(kms[0]) == 0x27EFF04CCCFDAE09
    assert O.minimizer_hash(kms[0], 21, 15) == 0x248300182C14CDB3


def test_two_word_kmer_k51():
    s = "ACGTTGCATGCATGCCGATTACGTAGCTAGCTAGCTAGCAAGGTTCCAGTC"
    w = O.pack_kmer(s)
    assert [int(x) for x in w] == [0x1BE4E4E58F1B2727, 0x27242BD4B4000000]
    assert [int(x) for x in O.revcomp(w, 51)] == [0x87A05F9C9C9C9C6C, 0x369393906C000000]
    assert O.kmer_hash(w) == 0x13B6161F7916CA8C
    assert O.minimizer_hash(w, 51, 27) == 0x4B08A00C5E9620AE


def test_scalar_hash_known_answers():
    L = O.lib()
    assert L.orc_quick_hash(0) == 0x7B439D0C1FD00DE3
    assert L.orc_quick_hash(1) == 0xBEA952A971BA8E83
    one = np.array([1], dtype=np.uint64)
    assert L.orc_murmur3_x64_64(one.ctypes.data, 8) == 0x3B35D9502FC3EEAD


def test_minimizer_len_rule():
    # kmer_dht.cpp:117-119
    assert [O.lib().orc_minimizer_len(k) for k in (21, 33, 51, 55, 77)] == [15, 23, 27, 27, 27]


def test_num_longs_rule():
    # main.cpp:169-190: MAX_K = (k/32+1)*32
    assert [O.num_longs(k) for k in (21, 31, 32, 33, 63, 64, 77, 99)] == [1, 1, 2, 2, 2, 3, 3, 4]


def test_dmin_double_truncation():
    # (int)((1.0 - 0.9) * count): the product lands just below the integer at multiples of 10
    def dmin(count):
        # top count exactly at the threshold passes, one below fails
        for d in range(1, 7000):
            if O.get_ext([d, 0, 0, 0], count) != "X":
                return d
    assert [dmin(c) for c in (2, 19, 30, 40, 50, 100, 65535)] == [2, 2, 2, 3, 4, 9, 6553]


def test_ext_vote_ties_and_forks():
    assert O.get_ext([5, 0, 0, 0], 5) == "A"
    assert O.get_ext([5, 1, 0, 0], 6) == "A"       # runner-up below dmin 2
    assert O.get_ext([5, 2, 0, 0], 7) == "F"
    assert O.get_ext([1, 1, 0, 0], 2) == "X"
    assert O.get_ext([0, 0, 0, 0], 2) == "X"
    assert O.get_ext([3, 0, 0, 3], 6) == "F"
    assert O.get_ext([2, 1, 1, 2], 6) == "F"       # tie at the top is a fork once both reach dmin
    assert O.get_ext([0, 1, 0, 1], 2, dmin_thres=1) == "F"
    assert O.get_ext([0, 1, 0, 0], 2, dmin_thres=1) == "C"
    # count 40 -> dmin 3: runner-up of 2 no longer forks
    assert O.get_ext([30, 2, 0, 0], 40) == "A"
    assert O.get_ext([30, 3, 0, 0], 40) == "F"


def test_rolling_get_kmers_matches_direct_pack():
    rng = np.random.default_rng(7)
    for k in (5, 21, 31, 32, 33, 51, 64, 77, 96, 99):
        s = "".join(rng.choice(list("ACGTN"), size=k + 40))
        kms = O.get_kmers(s, k)
        assert len(kms) == 41
        for i in range(41):
            assert (kms[i] == O.pack_kmer(s[i:i + k])).all(), (k, i)


def test_revcomp_is_involution_and_matches_strings():
    rng = np.random.default_rng(8)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for k in (3, 21, 31, 32, 33, 51, 63, 64, 65, 77, 127):
        for _ in range(20):
            s = "".join(rng.choice(list("ACGT"), size=k))
            w = O.pack_kmer(s)
            rc = O.revcomp(w, k)
            assert (rc == O.pack_kmer("".join(comp[c] for c in reversed(s)))).all(), (k, s)
            assert (O.revcomp(rc, k) == w).all()


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference at build time)")
def test_hashes_match_reference_build():
    ref = C.CDLL(REF_SO)
    ref.MurmurHash3_x64_64.restype = C.c_uint64
    ref.MurmurHash3_x64_64.argtypes = [C.c_void_p, C.c_uint32]
    ref.quick_hash.restype = C.c_uint64
    ref.quick_hash.argtypes = [C.c_uint64]
    L = O.lib()
    rng = np.random.default_rng(9)
    for n in list(range(0, 40)) + [64, 100, 255]:
        buf = rng.integers(0, 256, size=max(n, 1), dtype=np.uint8)
        assert L.orc_murmur3_x64_64(buf.ctypes.data, n) == ref.MurmurHash3_x64_64(buf.ctypes.data, n), n
    for v in [0, 1, 2**63, 2**64 - 1] + [int(x) for x in rng.integers(0, 2**63, size=200)]:
        assert L.orc_quick_hash(v) == ref.quick_hash(v)
