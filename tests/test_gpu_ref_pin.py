"""S3 / S4 / the slot hash pinned to the reference's OWN device code.

oracle/_ref/libref_gpu_common.so is oracle/ref_gpu_common.hip, which #includes -- by path, unmodified -- the
reference's src/gpu-utils/gpu_common.hpp (pack_seq_to_kmer, revcomp, comp_nucleotide) and
src/kcount/kcount-gpu/gpu_hash_funcs.hpp (gpu_murmurhash3_64); `make -C oracle ref` builds it where the reference
checkout exists and the binary travels to the GPU box.  Compared here, bit for bit, on ACGT-only input (the reference's
GPU twin rejects every other character):

  * the oracle's orc_pack_kmer / orc_revcomp / orc_kmer_hash (the restatement of src/kmer.cpp:155-262,490-510,470-473),
  * the HIP path's canonical records: every entry of kc_dump_table is min(k-mer, reverse complement) of the
    reference's own arithmetic with the multiplicity the reference's windows give it,
  * the complement the extension swap uses (utils.cpp:132-159 on the CPU side, comp_nucleotide on the GPU side).

S5-S9 (extension bookkeeping, saturation, vote, purge) stay pinned by hand cases and the second restatement only.
"""
import ctypes as C
import os

import numpy as np
import pytest

import mhm2_kmer_analysis_v2_amd as pkg
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu

REF_SO = os.path.join(os.path.dirname(O.__file__), "_ref", "libref_gpu_common.so")
needs_ref = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libref_gpu_common.so not built (needs /root/reference at build time)")

KS = [21, 31, 32, 33, 51, 63, 77]


def ref_lib():
    L = C.CDLL(REF_SO)
    L.ref_gpu_kmers.restype = C.c_int
    L.ref_gpu_kmers.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
    L.ref_gpu_comp.restype = C.c_int
    L.ref_gpu_comp.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
    return L


def ref_kmers(seq, k):
    """(k-mers, reverse complements, hashes, hashes of the reverse complements, ok) of every window, by the reference."""
    nl = k // 32 + 1
    npos = len(seq) - k + 1
    km = np.zeros((npos, nl), dtype=np.uint64)
    rc = np.zeros((npos, nl), dtype=np.uint64)
    h = np.zeros(npos, dtype=np.uint64)
    rh = np.zeros(npos, dtype=np.uint64)
    ok = np.zeros(npos, dtype=np.uint8)
    r = ref_lib().ref_gpu_kmers(seq.encode(), len(seq), k, nl, km.ctypes.data, rc.ctypes.data, h.ctypes.data, rh.ctypes.data, ok.ctypes.data)
    assert r == 0, "ref_gpu_kmers: HIP error %d" % r
    return km, rc, h, rh, ok


def rand_seq(rng, n, lower=0.0):
    s = "".join(rng.choice(list("ACGT"), size=n))
    if lower:
        s = "".join(c.lower() if rng.random() < lower else c for c in s)
    return s


@needs_ref
@pytest.mark.parametrize("k", KS)
def test_oracle_primitives_match_the_reference_device_code(k):
    rng = np.random.default_rng(4000 + k)
    L = O.lib()
    nl = k // 32 + 1
    # random sequence, a low-complexity stretch, mixed case (pack_seq_to_kmer upper-cases)
    seq = rand_seq(rng, 700, lower=0.3) + "A" * (k + 5) + "ACGT" * (k // 2 + 3) + "T" * (k + 2) + rand_seq(rng, 300)
    km, rc, h, rh, ok = ref_kmers(seq, k)
    assert ok.all()
    up = seq.upper().encode()
    u64p = C.POINTER(C.c_uint64)
    for i in range(len(seq) - k + 1):
        w = np.zeros(nl, dtype=np.uint64)
        L.orc_pack_kmer(up[i:i + k], k, w.ctypes.data_as(u64p))
        assert (w == km[i]).all(), "pack at %d" % i
        r = np.zeros(nl, dtype=np.uint64)
        L.orc_revcomp(w.ctypes.data_as(u64p), k, r.ctypes.data_as(u64p))
        assert (r == rc[i]).all(), "revcomp at %d" % i
        assert L.orc_kmer_hash(w.ctypes.data_as(u64p), nl) == int(h[i]), "hash at %d" % i
        assert L.orc_kmer_hash(r.ctypes.data_as(u64p), nl) == int(rh[i]), "hash of the reverse complement at %d" % i
    # all k-mers of a read at once (Kmer::get_kmers' shifting form) against the reference's per-window packing
    out = np.zeros((len(seq) - k + 1, nl), dtype=np.uint64)
    n = L.orc_get_kmers(up, len(up), k, out.ctypes.data_as(u64p))
    assert n == len(seq) - k + 1 and (out == km).all()


@needs_ref
def test_reference_rejects_what_it_should_and_complements_like_the_oracle_swap():
    k = 21
    seq = "ACGTACGTACGTACGTACGTANCGTACGTACGTACGTACGTACG"
    _, _, _, _, ok = ref_kmers(seq, k)
    npos = len(seq) - k + 1
    want = np.array([0 if "N" in seq[i:i + k] else 1 for i in range(npos)], dtype=np.uint8)
    assert (ok == want).all()  # F4a: the GPU twin drops windows with N; the CPU contract maps N -> G (pinned by the survey's KATs)
    chars = b"ACGTN0"
    out = C.create_string_buffer(len(chars))
    assert ref_lib().ref_gpu_comp(chars, len(chars), out) == 0
    assert out.raw == b"TGCAN0"  # what S5's swap complements with (kcount_cpu.cpp:328-333, utils.cpp:132-159)


@needs_ref
@pytest.mark.parametrize("tuning", [None, dict(writers=3, p1=256, p2=256, slots=512), dict(mode=1)], ids=["bucketed", "compact-or-small", "table"])
@pytest.mark.parametrize("k", KS)
def test_hip_path_canonical_records_match_the_reference_device_code(k, tuning):
    """Every entry the HIP path holds before the purge = min(k-mer, rc) by the reference's pack_seq_to_kmer / revcomp,
    counted once per window that has both neighbours (S5), over ACGT-only reads."""
    rng = np.random.default_rng(5000 + k)
    nl = k // 32 + 1
    genome = rand_seq(rng, 2500)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    reads = []
    for _ in range(400):
        ln = int(rng.integers(k + 2, k + 120))
        st = int(rng.integers(0, len(genome) - ln))
        s = genome[st:st + ln]
        if rng.random() < 0.5:
            s = "".join(comp[c] for c in reversed(s))
        reads.append(s)
    quals = ["I" * len(r) for r in reads]
    want = {}
    for r in reads:
        km, rc, _, _, ok = ref_kmers(r, k)
        assert ok.all()
        for i in range(1, len(r) - k):  # windows with a left and a right neighbour (kcount_cpu.cpp:320)
            a, b = tuple(int(x) for x in km[i]), tuple(int(x) for x in rc[i])
            c = min(a, b)  # words compared in order, unsigned (kmer.cpp:270-277)
            want[c] = want.get(c, 0) + 1
    b, q, offs = O.reads_to_arrays(reads, quals)
    with pkg.KmerCounter(k, tuning=tuning) as kc:
        kc.submit_reads(b, q, offs)
        kc.flush()
        keys, counts, _ = kc.dump_table()
    got = {tuple(int(x) for x in keys[i]): int(counts[i]) for i in range(len(counts))}
    assert len(got) == len(counts)
    assert got == want
    assert keys.shape[1] == nl
