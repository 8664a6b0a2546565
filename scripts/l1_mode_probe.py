"""Level 1 runs in one of a few modes (31.8 / 36.5 / 39.7 ms at 50 M reads) that stay fixed for a context: does a probe of
the fresh arena (KC_ARENA_PROBE: level 1's write pattern alone) tell them apart?  Several contexts in ONE process, two
alive at a time so that successive arenas get different physical memory."""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["KC_ARENA_PROBE"] = "1"

import torch
import mhm2_kmer_analysis_v2_amd as pkg
n = 50_000_000; k, L = 21, 150
dev = torch.device("cuda", 0)
b = torch.empty(n * L, dtype=torch.uint8, device=dev); q = torch.empty_like(b); o = torch.empty(n + 1, dtype=torch.int64, device=dev)
est = int(64 * 4_000_000 + n * L * 0.005 * k * 1.05) + (1 << 20)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
prev = None
for trial in range(3):
    kc = pkg.KmerCounter(k, device=0, max_elems=est, time_kernels=True, max_kmers_buffered=int(n * (L - k - 1) * 1.02) + (1 << 20))
    kc.set_stream(s.cuda_stream)
    if trial == 0:
        kc.synth_reads_device(b, q, o, n, first_read=0, params=pkg.synth_params())
    ts = []
    for it in range(2):
        kc.reset(); kc.kernel_times(clear=True)
        kc.submit_reads(b, q, o, nreads=n); torch.cuda.synchronize()
        kt = kc.kernel_times()
        ts.append((round(kt["kc_l1_reads_kernel"][1], 2)))
    print("context", trial, "l1 ms", ts, flush=True)
    if prev is not None:
        prev.close()
    prev = kc
os._exit(0)
