"""k-mers/s when the reads start in host memory (kc_submit_reads with on_device=0): the PCIe-inclusive figure."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import mhm2_kmer_analysis_v2_amd as pkg
n, L, k = 10_000_000, 150, 21
db = torch.empty(n * L, dtype=torch.uint8, device="cuda"); dq = torch.empty_like(db); do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
with pkg.KmerCounter(k, max_elems=500_000_000, max_kmers_buffered=n * 130) as kc:
    kc.synth_reads_device(db, dq, do, n)
    hb, hq, ho = db.cpu().numpy(), dq.cpu().numpy(), do.cpu().numpy().astype("uint64")
    for it in range(3):
        kc.reset()
        t0 = time.perf_counter(); kc.submit_reads(hb, hq, ho); kc.finalize(); dt = time.perf_counter() - t0
        print("host-resident input: %.1f ms, %.2f G k-mers/s, %.1f GB/s of input over PCIe" % (dt * 1e3, n * (L - k + 1) / dt / 1e9, 2 * n * L / dt / 1e9))
    for it in range(2):
        kc.reset()
        t0 = time.perf_counter(); kc.submit_reads(db, dq, do, nreads=n); kc.finalize(); dt = time.perf_counter() - t0
        print("HBM-resident input:  %.1f ms, %.2f G k-mers/s" % (dt * 1e3, n * (L - k + 1) / dt / 1e9))
