import os, sys
sys.path.insert(0, os.getcwd())
import torch
import mhm2_kmer_analysis_v2_amd as pkg
n = 50_000_000; k, L = 21, 150
dev = torch.device("cuda", 0)
pad = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if pad >= 0:
    buf = torch.empty(2 * n * L + pad + 4096, dtype=torch.uint8, device=dev)
    b = buf[:n * L]; q = buf[n * L + pad: 2 * n * L + pad]
else:
    b = torch.empty(n * L, dtype=torch.uint8, device=dev); q = torch.empty_like(b)
o = torch.empty(n + 1, dtype=torch.int64, device=dev)
kc = pkg.KmerCounter(k, device=0, max_elems=int(64*4_000_000 + n*L*0.005*k*1.05) + (1 << 20), time_kernels=True, max_kmers_buffered=int(n * (L - k - 1) * 1.02) + (1 << 20))
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s); kc.set_stream(s.cuda_stream)
kc.synth_reads_device(b, q, o, n, first_read=0, params=pkg.synth_params())
ts = []
for it in range(3):
    kc.reset(); kc.kernel_times(clear=True)
    kc.submit_reads(b, q, o, nreads=n); torch.cuda.synchronize()
    ts.append(round(kc.kernel_times()["kc_l1_reads_kernel"][1], 2))
print("pad", pad, "bases %x quals %x diff %x offs %x" % (b.data_ptr(), q.data_ptr(), q.data_ptr() - b.data_ptr(), o.data_ptr()), ts, flush=True)
os._exit(0)
