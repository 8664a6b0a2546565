"""The reference's wire format: how fast are the sender's block call (kc_build_supermers =
ParseAndPackGPUDriver::process_seq_block + pack_seq_block, parse_and_pack.cpp:281-336) and the receiver's
(kc_submit_packed_supermers = insert_supermer_block, gpu_hash_table.cpp:655-695)?  A '_'-joined, case-masked block of
synthetic 150-base reads (the generator of bench.py), at the reference's block size (just under its 3 MB limit,
parse_and_pack.cpp:281-286) and at larger ones; GB/s of block characters, whole call (host string in, host results
out), best of three.  python scripts/wire_rate_probe.py [k] [ranks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mhm2_kmer_analysis_v2_amd as pkg

k = int(sys.argv[1]) if len(sys.argv) > 1 else 21
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = 150
nmax = (512 << 20) // (L + 1)
bases, quals, _ = pkg.synth_reads_host(nmax)
blk = np.full((nmax, L + 1), ord("_"), dtype=np.uint8)
b = bases.reshape(nmax, L)
blk[:, :L] = np.where(quals.reshape(nmax, L) < 33 + 20, b | 0x20, b)  # low quality = lower case (kcount.cpp:81)
blk = blk.reshape(-1)
del bases, quals, b

import ctypes as C
sender = pkg.KmerCounter(k, device=0, max_elems=1 << 20, rank_me=0, rank_n=ranks, reference_owner=True)
print("k=%d, %d ranks; block of %d-base reads joined by '_'; host buffers pinned" % (k, ranks, L))
hblk = torch.from_numpy(blk).pin_memory()
cap = len(blk) // 4
hout = torch.empty(cap * 12, dtype=torch.uint8).pin_memory()
hpacked = torch.empty((len(blk) + 1) // 2, dtype=torch.uint8).pin_memory()
for mb in (2.9, 32, 256, 512):
    n = int(mb * (1 << 20)) // (L + 1)
    nb = n * (L + 1)
    best = 1e9
    ns, nk = C.c_uint32(0), C.c_uint32(0)
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = pkg.lib().kc_build_supermers(sender._h, hblk.data_ptr(), nb, 0, hout.data_ptr(), cap, C.byref(ns), C.byref(nk), hpacked.data_ptr())
        best = min(best, time.perf_counter() - t0)
        assert rc == 0, rc
    print("kc_build_supermers  %7.1f MB block: %8.2f ms = %6.2f GB/s  (%d supermers, %d k-mers, %.2f k-mers per supermer)" %
          (nb / 1e6, best * 1e3, nb / best / 1e9, ns.value, nk.value, nk.value / max(1, ns.value)), flush=True)

# receiver: the sender's packed block as a whole stands for a target's buffer (valid nibbles, a separator between reads;
# for the rate it does not matter where the supermers were cut)
n = (256 << 20) // (L + 1)
block = blk[: n * (L + 1)]
buf = torch.from_numpy(sender.build_supermers(block, capacity=len(block) // 4)[4]).pin_memory().numpy()
recv = pkg.KmerCounter(k, device=0, max_elems=1 << 26, max_kmers_buffered=len(buf) * 2 + (1 << 20))  # the bucketed path, as in bench.py
for mb in (1, 16, len(buf) / (1 << 20)):
    m = int(mb * (1 << 20))
    piece = buf[:m]
    best = 1e9
    for it in range(3):
        recv.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        recv.submit_packed_supermers(piece)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("kc_submit_packed_supermers %6.1f MB packed (%.1f M characters): %8.2f ms = %6.2f GB/s packed" %
          (m / 1e6, 2 * m / 1e6, best * 1e3, m / best / 1e9), flush=True)
del sender, recv
