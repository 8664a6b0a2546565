"""Host-resident packed reads: how the end-to-end time depends on the copy block size (KC_HOST_BLOCK), and what the
kernels take meanwhile.  python scripts/host_pipe_probe.py [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mhm2_kmer_analysis_v2_amd as pkg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
k, L = 21, 150
dev = torch.device("cuda", 0)
b = torch.empty(n * L, dtype=torch.uint8, device=dev); q = torch.empty_like(b); o = torch.empty(n + 1, dtype=torch.int64, device=dev)
est = int(64 * 4_000_000 + n * L * 0.005 * k * 1.05) + (1 << 20)
kc = pkg.KmerCounter(k, device=0, max_elems=est, time_kernels=True, max_kmers_buffered=int(n * (L - k - 1) * 1.02) + (1 << 20))
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s); kc.set_stream(s.cuda_stream)
kc.synth_reads_device(b, q, o, n, first_read=0, params=pkg.synth_params())
code = torch.full((256,), 4, dtype=torch.uint8, device=dev)
for i, ch in enumerate(b"ACGT"):
    code[ch] = i
packed = torch.empty_like(b)
step = 1 << 28
for at in range(0, b.numel(), step):
    packed[at:at + step] = code[b[at:at + step].long()] | ((q[at:at + step].to(torch.int16) - 33).clamp(max=31).to(torch.uint8) << 3)
hp = packed.cpu().pin_memory(); ho = o.cpu().numpy().astype(np.uint64)
del packed, b, q
for blk in (64 << 20, 256 << 20, 1 << 30, 8 << 30):
    os.environ["KC_HOST_BLOCK"] = str(blk)
    for it in range(2):
        kc.reset(); kc.kernel_times(clear=True); torch.cuda.synchronize()
        t0 = time.perf_counter(); kc.submit_packed_reads(hp.numpy(), ho); torch.cuda.synchronize(); t1 = time.perf_counter()
        kc.finalize(); torch.cuda.synchronize(); t2 = time.perf_counter()
    kt = {k_: round(v[1], 1) for k_, v in kc.kernel_times().items()}
    print("block %5d MiB: submit %.1f ms (%.1f GB/s), finalize %.1f ms, total %.1f ms; kernels %s" % (blk >> 20, (t1 - t0) * 1e3, n * L / (t1 - t0) / 1e9, (t2 - t1) * 1e3, (t2 - t0) * 1e3, kt), flush=True)
os._exit(0)
