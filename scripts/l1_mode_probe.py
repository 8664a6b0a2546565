"""Level 1 runs in one of two modes (31.8 / 39.7 ms at 50 M reads) that stay fixed within a process: does the mode follow
the arena allocation?  Several contexts, one after the other, in ONE process; optionally a dummy allocation in between."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import mhm2_kmer_analysis_v2_amd as pkg
n = 50_000_000; k, L = 21, 150
dev = torch.device("cuda", 0)
b = torch.empty(n * L, dtype=torch.uint8, device=dev); q = torch.empty_like(b); o = torch.empty(n + 1, dtype=torch.int64, device=dev)
est = int(64 * 4_000_000 + n * L * 0.005 * k * 1.05) + (1 << 20)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
dummies = []
for trial in range(6):
    kc = pkg.KmerCounter(k, device=0, max_elems=est, time_kernels=True, max_kmers_buffered=int(n * (L - k - 1) * 1.02) + (1 << 20))
    kc.set_stream(s.cuda_stream)
    if trial == 0:
        kc.synth_reads_device(b, q, o, n, first_read=0, params=pkg.synth_params())
    ts = []
    for it in range(2):
        kc.reset(); kc.kernel_times(clear=True)
        kc.submit_reads(b, q, o, nreads=n); torch.cuda.synchronize()
        ts.append(round(kc.kernel_times()["kc_l1_reads_kernel"][1], 2))
    print("context", trial, "dummies", len(dummies), ts, flush=True)
    kc.close()
    if trial % 2 == 1:
        dummies.append(torch.empty(int(3e9) + trial * 12345678, dtype=torch.uint8, device=dev))  # shifts the next arena
os._exit(0)
