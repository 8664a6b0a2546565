"""Does a pinned H2D copy overlap the level-1 kernel?  The copy alone, level 1 alone (input in HBM), both at once."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mhm2_kmer_analysis_v2_amd as pkg
n, k, L = 50_000_000, 21, 150
dev = torch.device("cuda", 0)
b = torch.empty(n * L, dtype=torch.uint8, device=dev); q = torch.empty_like(b); o = torch.empty(n + 1, dtype=torch.int64, device=dev)
kc = pkg.KmerCounter(k, device=0, max_elems=1 << 20, time_kernels=True, max_kmers_buffered=int(n * (L - k - 1) * 1.02) + (1 << 20))
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s); kc.set_stream(s.cuda_stream)
kc.synth_reads_device(b, q, o, n, first_read=0, params=pkg.synth_params())
h = torch.empty(7_500_000_000, dtype=torch.uint8).pin_memory()
d = torch.empty_like(h, device=dev)
cs = torch.cuda.Stream(device=dev)
def copy():
    with torch.cuda.stream(cs):
        d.copy_(h, non_blocking=True)
def l1(times=4):
    for _ in range(times):
        kc.reset(); kc.submit_reads(b, q, o, nreads=n)
for name, fn in (("copy alone", lambda: copy()), ("4 x level 1 alone", lambda: l1()), ("both", lambda: (copy(), l1()))):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-18s %.1f ms" % (name, dt * 1e3), flush=True)
os._exit(0)
