"""Host-resident reads through the overlapped staging of kc_submit_reads / kc_submit_packed_reads (two device slots,
two pinned host slots, copies on their own stream): many blocks, pageable and pinned sources, against the oracle.
Role in the reference: the block loop of src/kcount/kcount_gpu.cpp:110-165."""
import os

import numpy as np
import pytest

import mhm2_kmer_analysis_v2_amd as pkg
from helpers import random_reads
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture
def small_blocks(monkeypatch):
    monkeypatch.setenv("KC_HOST_BLOCK", str(1 << 20))  # 1 MiB of bases per block: a few MB of reads make many blocks


def _input(k, n=75000, seed=5):
    rng = np.random.default_rng(seed)
    reads, quals = random_reads(rng, n, min_len=k - 3, max_len=k + 130, genome_len=40000, err=0.01)
    b, q, offs = O.reads_to_arrays(reads, quals)
    assert len(b) > 5 * (1 << 20)  # at least five blocks
    o = O.Oracle(k, nranks=8, nthreads=8)
    o.add_reads(b, q, offs)
    want = o.finalize()
    st = o.stats()
    o.close()
    return b, q, offs, want, st


@pytest.mark.parametrize("k", [21, 51])
@pytest.mark.parametrize("source", ["pageable", "pinned"])
def test_host_reads_in_many_blocks(small_blocks, k, source):
    import torch
    b, q, offs, want, wst = _input(k)
    keep = None
    if source == "pinned":
        keep = (torch.from_numpy(b).pin_memory(), torch.from_numpy(q).pin_memory())
        b, q = keep[0].numpy(), keep[1].numpy()
    with pkg.KmerCounter(k) as kc:
        kc.submit_reads(b, q, offs)
        got = kc.sorted_results()
        st = kc.stats()
    for g, w in zip(got, want):
        assert g.shape == w.shape and (g == w).all()
    assert st["num_reads"] == wst["reads"] and st["raw_kmers"] == wst["raw_kmers"] and st["num_bases"] == len(b)


def test_host_packed_reads_in_many_blocks(small_blocks):
    k = 21
    b, q, offs, want, _ = _input(k, seed=6)
    # the read cache's byte: base code 0-4 (ACGTN) | min(quality - offset, 31) << 3 (src/packed_reads.cpp:99-126)
    code = np.full(256, 4, dtype=np.uint8)
    for i, ch in enumerate("ACGT"):
        code[ord(ch)] = i
        code[ord(ch.lower())] = i
    packed = code[b] | (np.minimum(q.astype(np.int32) - 33, 31).astype(np.uint8) << 3)
    with pkg.KmerCounter(k) as kc:
        kc.submit_packed_reads(packed, offs)
        got = kc.sorted_results()
    for g, w in zip(got, want):
        assert g.shape == w.shape and (g == w).all()


def test_blocks_of_one_long_read(small_blocks):
    """a read longer than a block gets a block of its own"""
    k = 33
    rng = np.random.default_rng(8)
    long_read = "".join(rng.choice(list("ACGT"), size=(3 << 20) + 17))
    reads = ["ACGT" * 30, long_read, "TTGACCA" * 20, long_read[1000:2000]]
    b, q, offs = O.reads_to_arrays(reads)
    o = O.Oracle(k, nranks=2, nthreads=2)
    o.add_reads(b, q, offs)
    want = o.finalize()
    o.close()
    with pkg.KmerCounter(k) as kc:
        kc.submit_reads(b, q, offs)
        got = kc.sorted_results()
    for g, w in zip(got, want):
        assert g.shape == w.shape and (g == w).all()


def test_host_blocks_through_a_small_buffer_and_mixed_sources(small_blocks):
    """The host pipe counts a block's occurrences itself instead of asking the device (no wait per block): the count must
    stay right across spill passes (a buffer a quarter of the input) and when device-resident blocks, whose count only the
    device knows, come in between."""
    import torch
    k = 21
    b, q, offs, want, wst = _input(k, seed=9)
    n = len(offs) - 1
    cut1, cut2 = n // 3, 2 * n // 3

    def part(lo, hi):
        o = offs[lo:hi + 1] - offs[lo]
        return b[offs[lo]:offs[hi]], q[offs[lo]:offs[hi]], o.astype(np.uint64)

    occ = int(np.maximum(np.diff(offs.astype(np.int64)) - k - 1, 0).sum())
    with pkg.KmerCounter(k, max_kmers_buffered=occ // 4) as kc:
        kc.submit_reads(*part(0, cut1))                      # host: counted on the host
        pb, pq, po = part(cut1, cut2)                        # device-resident: counted on the device
        dev = torch.device("cuda", 0)
        db, dq, do = torch.from_numpy(pb.copy()).to(dev), torch.from_numpy(pq.copy()).to(dev), torch.from_numpy(po.astype(np.int64)).to(dev)
        kc.submit_reads(db, dq, do)                          # (the tensors stay alive until the results are out)
        kc.submit_reads(*part(cut2, n))                      # host again: picks the device's count up first
        got = kc.sorted_results()
        st = kc.stats()
    for g, w in zip(got, want):
        assert g.shape == w.shape and (g == w).all()
    assert st["raw_kmers"] == wst["raw_kmers"]


@pytest.mark.parametrize("k,tuning", [
    (21, None),
    (21, dict(writers=3, p1=8, p2=16, slots=2048, chunk1=16, chunk2=16)),                                     # many chunks per chain
    (21, dict(writers=2, p1=2, p2=4, slots=4096, chunk2=16, chain2_max=10, ovf_capacity=20000)),             # region chains overflow; the second list grows
    (51, dict(writers=4, p1=8, p2=8, slots=128, chunk1=16, chain1_max=6, chunk2=16, chain2_max=12, ovf_capacity=1 << 20)),  # everything at once, wide records
], ids=["default", "many-chunks", "chain2-overflow-retry", "everything-wide"])
def test_level_2_in_instalments_equals_level_2_at_once(small_blocks, monkeypatch, k, tuning):
    """The host pipe runs level 2 over what has arrived after every block but the last (region chains continue, every
    bucket's part of the level-2 arena is fixed before its records are known); with KC_L2_INSTALMENTS=0 it runs once
    at the end as before.  Both must give the oracle's set -- also where regions overflow their chains or their
    bucket's part, and where the second overflow list has to grow (the whole pass is then run again at once)."""
    b, q, offs, want, wst = _input(k, seed=11)
    got = {}
    for inst in ("1", "0"):
        monkeypatch.setenv("KC_L2_INSTALMENTS", inst)
        with pkg.KmerCounter(k, tuning=tuning) as kc:
            kc.submit_reads(b, q, offs)
            got[inst] = kc.sorted_results()
            st = kc.stats()
        for g, w in zip(got[inst], want):
            assert g.shape == w.shape and (g == w).all()
        assert st["num_unique"] == wst["unique"] and st["sum_counts"] == wst["sum_counts"]


@pytest.mark.parametrize("shift", [1, 7, 16, 100])
def test_pinned_arrays_of_different_alignment_keep_the_wide_loads(small_blocks, shift):
    """Bases and qualities in two pinned buffers whose starts differ modulo 16: every block's copy starts on a page of ITS
    source, so the two leads differ modulo 16 -- the qualities are then placed up to 15 bytes further into their device
    slot, a read's bases and qualities stay co-aligned, and level 1 runs its 16-byte-load instantiation for every block
    (it used to fall back, silently, to byte-loaded qualities)."""
    import torch
    k = 21
    b, q, offs, want, _ = _input(k, seed=8)
    pb = torch.empty(len(b) + 4096, dtype=torch.uint8).pin_memory()
    pq = torch.empty(len(q) + 4096, dtype=torch.uint8).pin_memory()
    bb = pb.numpy()[:len(b)]
    qq = pq.numpy()[shift:shift + len(q)]
    bb[:] = b
    qq[:] = q
    assert (bb.ctypes.data - qq.ctypes.data) % 16 == (-shift) % 16
    with pkg.KmerCounter(k, time_kernels=True) as kc:
        kc.submit_reads(bb, qq, offs)
        got = kc.sorted_results()
        names = set(kc.kernel_times())
    for g, w in zip(got, want):
        assert g.shape == w.shape and (g == w).all()
    assert any(n.startswith("kc_l1_reads") for n in names)
    assert not any("byte-loaded" in n for n in names), names
