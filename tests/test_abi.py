"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol the
header declares, and its host-only entry points behave (no compute without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mhm2_kmer_analysis_v2_amd as pkg
from mhm2_kmer_analysis_v2_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "kcount_mi355.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kc_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = pkg.lib()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libkcount_mi355.so does not export %s" % n
        assert n in _lib.SYMBOLS, "python binding lacks %s" % n
    assert sorted(_lib.SYMBOLS) == names


def test_version_and_error_strings():
    L = pkg.lib()
    assert L.kc_abi_version() == 1
    assert L.kc_error_string(0) == b"ok"
    assert b"k-mer" in L.kc_error_string(-2)


def test_num_longs_matches_reference_rule():
    L = pkg.lib()
    assert [L.kc_num_longs(k) for k in (21, 31, 32, 33, 63, 64, 77, 99)] == [1, 1, 2, 2, 2, 3, 3, 4]
    assert [L.kc_record_longs(k) for k in (21, 29, 30, 31, 32, 62, 63, 95, 125)] == [1, 1, 2, 2, 2, 3, 3, 4, 4]


def test_create_argument_checks_need_no_gpu():
    L = pkg.lib()
    st = C.c_int(0)
    for k, want in ((2, -2), (128, -2), (126, -2), (127, -2)):  # 126, 127: a fifth record word would be needed
        cfg = _lib.kc_config(kmer_len=k, qual_offset=33, dmin_thres=2, device=0, rank_me=0, rank_n=1)
        assert not L.kc_create(C.byref(cfg), C.byref(st)) and st.value == want, k
    cfg = _lib.kc_config(kmer_len=21, qual_offset=33, dmin_thres=2, device=0, rank_me=2, rank_n=2)
    assert not L.kc_create(C.byref(cfg), C.byref(st)) and st.value == -1
    assert not L.kc_create(None, C.byref(st)) and st.value == -1


def test_no_gpu_fails_loudly():
    if pkg.lib().kc_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.KcError) as e:
        pkg.KmerCounter(21)
    assert e.value.status == -3  # KC_ERR_NO_DEVICE: there is no CPU fallback


def test_owner_is_deterministic_and_balanced():
    L = pkg.lib()
    rng = np.random.default_rng(3)
    keys = rng.integers(0, 2**63, size=4000, dtype=np.uint64) << np.uint64(1)
    keys &= ~np.uint64(0x3F) & np.uint64(0xFFFFFFFFFFFFFFFF)
    keys &= np.uint64(0xFFFFFFFFFFC00000)  # k=21 uses the top 42 bits
    owners = [L.kc_owner(keys[i:i + 1].ctypes.data, 21, 8) for i in range(len(keys))]
    assert owners == [L.kc_owner(keys[i:i + 1].ctypes.data, 21, 8) for i in range(len(keys))]
    hist = np.bincount(owners, minlength=8)
    assert hist.min() > 400 and hist.max() < 600
    assert all(L.kc_owner(keys[i:i + 1].ctypes.data, 21, 1) == 0 for i in range(10))


def test_host_synth_reads_shape_and_determinism():
    p = pkg.synth_params(num_genomes=3, min_genome_len=500, max_genome_len=900, read_len=50, n_rate=0.01)
    b1, q1, o1 = pkg.synth_reads_host(200, params=p)
    b2, q2, o2 = pkg.synth_reads_host(100, first_read=100, params=p)
    assert (o1 == np.arange(201) * 50).all()
    assert (b1[100 * 50:] == b2).all() and (q1[100 * 50:] == q2).all()  # a block is a slice of the stream
    assert set(bytes(b1)) <= set(b"ACGTN") and set(bytes(q1)) <= set(b"I#")
    err = (q1 == ord("#")).mean()
    assert 0.005 < err < 0.05


def test_reference_owner_matches_oracle_target_rank():
    """kc_owner_reference is the reference's get_kmer_target_rank (kmer_dht.cpp:192-196), whose restatement in
    the oracle is pinned by the SURVEY known answers for minimizer_hash_fast."""
    from oracle import cpu_oracle as O
    L = pkg.lib()
    rng = np.random.default_rng(4)
    for k in (21, 33, 51, 55, 77, 99):
        m = O.lib().orc_minimizer_len(k)
        for _ in range(200):
            s = "".join(rng.choice(list("ACGT"), size=k))
            w = O.pack_kmer(s)
            rc = O.revcomp(w, k)
            canon = rc if tuple(rc) < tuple(w) else w
            for n in (2, 8, 13):
                want = O.target_rank(canon, k, m, n)
                assert L.kc_owner_reference(canon.ctypes.data, k, n) == want
                assert L.kc_owner_reference(np.ascontiguousarray(w).ctypes.data, k, n) == want  # strand-independent
