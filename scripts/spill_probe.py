"""A buffer smaller than the input: the stage in several passes (bk_spill_pass), against the same input in one pass.
usage: spill_probe.py [reads] [fraction of the input the buffer holds] [blocks]"""
import sys
import time

import torch

import mhm2_kmer_analysis_v2_amd as pkg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 8
k, L = 21, 150
dev = torch.device("cuda", 0)
p = pkg.synth_params()
b = torch.empty(n * L, dtype=torch.uint8, device=dev)
q = torch.empty_like(b)
o = torch.empty(n + 1, dtype=torch.int64, device=dev)
est = int(64 * 4_000_000 + n * L * p.sub_error_rate * k * 1.05) + (1 << 20)
full = int(n * (L - k - 1) * 1.02) + (1 << 20)
for name, cap in (("one pass", full), ("buffer = %.2f of the input" % frac, int(full * frac))):
    with pkg.KmerCounter(k, max_elems=est, max_kmers_buffered=cap, time_kernels=True) as kc:
        if name == "one pass":
            kc.synth_reads_device(b, q, o, n, params=p)
        for rep in range(2):
            kc.reset()
            kc.kernel_times(clear=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            per = (n + blocks - 1) // blocks
            for r0 in range(0, n, per):
                r1 = min(n, r0 + per)
                kc.submit_reads(b[r0 * L:], q[r0 * L:], o[r0:r1 + 1] - o[r0], nreads=r1 - r0)
            kc.finalize()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        st = kc.stats()
        kt = kc.kernel_times()
        print("%s: %.1f ms = %.1f G k-mers/s, total_kmers %d unique %d | %s" % (
            name, dt * 1e3, n * (L - k + 1) / dt / 1e9, st["total_kmers"], st["num_unique"],
            ", ".join("%s x%d %.1f" % (a.replace("kc_", "").replace("_kernel", ""), v[0], v[1]) for a, v in sorted(kt.items(), key=lambda kv: -kv[1][1]))), flush=True)
