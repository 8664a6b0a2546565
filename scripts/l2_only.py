"""Time levels 1 and 2 alone (no count): python scripts/l2_only.py [reads]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mhm2_kmer_analysis_v2_amd as pkg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
k, L = 21, 150
dev = torch.device("cuda", 0)
b = torch.empty(n * L, dtype=torch.uint8, device=dev); q = torch.empty_like(b); o = torch.empty(n + 1, dtype=torch.int64, device=dev)
kc = pkg.KmerCounter(k, device=0, max_elems=1_100_000_000, time_kernels=True, max_kmers_buffered=int(n * (L - k - 1) * 1.02) + (1 << 20))
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s); kc.set_stream(s.cuda_stream)
kc.synth_reads_device(b, q, o, n, first_read=0, params=pkg.synth_params())
for it in range(2):
    kc.reset(); kc.kernel_times(clear=True)
    kc.submit_reads(b, q, o, nreads=n)
    try:
        kc.finalize()  # the experiment builds stop after level 2 (KC_ERR_STATE)
    except pkg.KcError:
        pass
    torch.cuda.synchronize()
    print({k_: round(v[1], 2) for k_, v in kc.kernel_times().items()}, flush=True)
os._exit(0)
