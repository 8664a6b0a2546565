"""The reference's wire format on the device (N1 of SURVEY.md 8f): supermers + 4-bit packed block from the sender
entry point, device-side unpack at the receiver.
  * kc_build_supermers == the oracle's SeqBlockInserter::process_seq (kcount_cpu.cpp:73-103), supermer for supermer,
    with targets = KmerDHT::get_kmer_target_rank (kmer_dht.cpp:192-196);
  * the bytes a host cuts out of the packed block as src/kcount/kcount_gpu.cpp:153-161 does, unpacked with the
    reference's nibble codes, are the supermer's characters;
  * those bytes, handed to the target's context through kc_submit_packed_supermers, give the oracle's result set."""
import numpy as np
import pytest

import mhm2_kmer_analysis_v2_amd as pkg
from helpers import random_reads
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu

TO_BASE = "_acgtACGTN"  # gpu_hash_table.cpp:270, parse_and_pack.cpp:196-213


def cut(packed, offset, ln):
    """the host loop of kcount_gpu.cpp:153-161: the supermer's bytes, odd nibbles masked"""
    plen = ln // 2 + (1 if (offset % 2 or ln % 2) else 0)
    seq = bytearray(packed[offset // 2: offset // 2 + plen].tobytes())
    if offset % 2:
        seq[0] &= 15
    if (offset + ln) % 2:
        seq[-1] &= 240
    return bytes(seq)


def unpack(seq):
    return "".join(TO_BASE[b >> 4] + TO_BASE[b & 15] for b in seq)


@pytest.mark.parametrize("k,nranks", [(21, 5), (33, 3), (51, 8), (77, 2)])
def test_supermers_and_round_trip(k, nranks):
    rng = np.random.default_rng(200 + k)
    reads, quals = random_reads(rng, 1200, min_len=k - 4, max_len=k + 110, genome_len=3000, n_rate=0.004)
    masked = ["".join(c.lower() if ord(x) < 33 + 20 else c for c, x in zip(r, q)) for r, q in zip(reads, quals)]
    block = ("_".join(masked) + "_").encode()
    starts = np.cumsum([0] + [len(m) + 1 for m in masked])
    # the oracle: every read's supermers, placed in the block
    o = O.Oracle(k, nranks=nranks, nthreads=2)
    want = set()
    for m, s0 in zip(masked, starts):
        for t, st, ln in o.supermers(m):
            want.add((t, int(s0) + st, ln))
    b, q, offs = O.reads_to_arrays(reads, quals)
    o.add_reads(b, q, offs)
    res = o.finalize()
    ost = o.stats()
    o.close()
    with pkg.KmerCounter(k, rank_me=0, rank_n=nranks) as kc:
        targets, offsets, lens, nvalid, packed = kc.build_supermers(block)
    got = set(zip(targets.tolist(), offsets.tolist(), lens.tolist()))
    assert got == want and len(got) == len(targets) == ost["supermers"]
    assert nvalid == ost["kmers_inserted"] == sum(ln - k - 1 for _, _, ln in got)
    # the packed block is the block in the reference's nibble codes
    assert unpack(packed.tobytes())[:len(block)] == block.decode().replace("n", "N")
    # cut like the host does, unpack like the receiver does, and insert at the target
    per_target = [bytearray() for _ in range(nranks)]
    for t, off, ln in got:
        seq = cut(packed, off, ln)
        assert unpack(seq).strip("_") == block[off:off + ln].decode().replace("n", "N")
        per_target[t] += seq + b"_"
    parts = []
    for t in range(nranks):
        with pkg.KmerCounter(k) as kc:
            if per_target[t]:
                kc.submit_packed_supermers(np.frombuffer(bytes(per_target[t]), dtype=np.uint8))
            parts.append(kc.sorted_results())
    nl = pkg.lib().kc_num_longs(k)
    keys = np.concatenate([p[0] for p in parts])
    order = np.lexsort([keys[:, j] for j in range(nl - 1, -1, -1)])
    union = tuple(np.concatenate([p[i] for p in parts])[order] for i in range(4))
    for g, w in zip(union, res):
        assert g.shape == w.shape and (g == w).all()


def test_bad_nibble_is_reported():
    with pkg.KmerCounter(21) as kc:
        with pytest.raises(pkg.KcError) as e:
            kc.submit_packed_supermers(np.frombuffer(bytes([0x5A, 0x5B, 0x55]), dtype=np.uint8))  # 0xA, 0xB: no such codes
        assert e.value.status == -7


@pytest.mark.parametrize("k", [21, 63])
def test_wire_kernels_on_odd_addresses_and_lengths(k):
    """The tile / sixteen-bytes-at-a-time kernels have an aligned fast path and a byte path: a device-resident block that
    starts at every offset 0..16 of an allocation, lengths that are no multiple of anything (also shorter than k, one
    character, and longer than one tile of 3840 positions), packed bytes unpacked from an odd address -- each against
    the oracle's supermers and against the host-resident call."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(900 + k)
    reads, quals = random_reads(rng, 90, min_len=k - 3, max_len=k + 140, genome_len=2500, n_rate=0.01)
    masked = ["".join(c.lower() if ord(x) < 33 + 20 else c for c, x in zip(r, q)) for r, q in zip(reads, quals)]
    whole = ("_".join(masked) + "_").encode()
    o = O.Oracle(k, nranks=7, nthreads=1)
    dev = torch.device("cuda", 0)
    with pkg.KmerCounter(k, rank_me=0, rank_n=7) as kc:
        for off, ln in [(0, len(whole)), (1, len(whole) - 1), (3, 5000), (7, 4097), (8, 3841), (13, 777), (16, k + 1), (5, k + 2), (2, 1), (9, 16)]:
            block = whole[off:off + ln]
            # what the oracle makes of the same characters: read by read (a cut may split a read: its pieces are reads)
            want = set()
            at = 0
            for piece in block.decode().split("_"):
                for t, st, l2 in o.supermers(piece):
                    want.add((t, at + st, l2))
                at += len(piece) + 1
            host = kc.build_supermers(block)
            buf = torch.zeros(len(whole) + 64, dtype=torch.uint8, device=dev)
            buf[off:off + ln] = torch.frombuffer(bytearray(block), dtype=torch.uint8).to(dev)
            torch.cuda.synchronize()  # (the library works on a stream of its own)
            cap = max(16, ln)
            out = np.zeros(cap, dtype=np.dtype([("target", np.int32), ("offset", np.int32), ("len", np.uint16), ("pad", np.uint16)]))
            packed = np.zeros((ln + 1) // 2, dtype=np.uint8)
            n, nk = C.c_uint32(0), C.c_uint32(0)
            rc = pkg.lib().kc_build_supermers(kc._h, buf.data_ptr() + off, ln, 1, out.ctypes.data, cap, C.byref(n), C.byref(nk), packed.ctypes.data)
            assert rc == 0
            got = set(zip(out["target"][:n.value].tolist(), out["offset"][:n.value].tolist(), out["len"][:n.value].tolist()))
            assert got == want == set(zip(host[0].tolist(), host[1].tolist(), host[2].tolist())), (off, ln)
            assert nk.value == host[3] == sum(l2 - k - 1 for _, _, l2 in want)
            assert (packed == host[4]).all()
            assert unpack(packed.tobytes())[:ln] == block.decode().replace("n", "N")
    o.close()
    # receiver: the same packed bytes from an even and from an odd device address give the same table
    packed_all = None
    with pkg.KmerCounter(k, rank_me=0, rank_n=7) as kc:
        packed_all = kc.build_supermers(whole)[4]
    res = []
    for shift in (0, 1, 5):
        with pkg.KmerCounter(k) as kc:
            d = torch.zeros(len(packed_all) + 16, dtype=torch.uint8, device=dev)
            d[shift:shift + len(packed_all)] = torch.from_numpy(packed_all).to(dev)
            torch.cuda.synchronize()
            kc.submit_packed_supermers(d[shift:shift + len(packed_all)])
            res.append(kc.sorted_results())
    b, q, offs = O.reads_to_arrays(reads, quals)
    o = O.Oracle(k, nranks=2, nthreads=2)
    o.add_reads(b, q, offs)
    want = o.finalize()
    o.close()
    for r in res:
        for g, w in zip(r, want):
            assert g.shape == w.shape and (g == w).all()
