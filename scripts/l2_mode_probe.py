"""Level 2 has a rare fast draw (23.6 ms instead of 26-27 per 50 M reads, whole process): is it a property of the
allocations?  One process, a fresh context (fresh arenas) per round, the same reads: level-2 time per context beside the
arenas' addresses (KC_DEBUG_ADDR=1 prints them on stderr).  python scripts/l2_mode_probe.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KC_DEBUG_ADDR"] = "1"
import torch
import mhm2_kmer_analysis_v2_amd as pkg
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n, k, L = 50_000_000, 21, 150
dev = torch.device("cuda", 0)
b = torch.empty(n * L, dtype=torch.uint8, device=dev); q = torch.empty_like(b); o = torch.empty(n + 1, dtype=torch.int64, device=dev)
est = int(64 * 4_000_000 + n * L * 0.005 * k * 1.05) + (1 << 20)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
gen = pkg.KmerCounter(k, device=0, max_elems=1 << 20)
gen.set_stream(s.cuda_stream)
gen.synth_reads_device(b, q, o, n, first_read=0, params=pkg.synth_params())
torch.cuda.synchronize(); gen.close()
for r in range(rounds):
    kc = pkg.KmerCounter(k, device=0, max_elems=est, time_kernels=True, max_kmers_buffered=int(n * (L - k - 1) * 1.02) + (1 << 20))
    kc.set_stream(s.cuda_stream)
    for it in range(2):
        kc.reset(); kc.kernel_times(clear=True)
        kc.submit_reads(b, q, o, nreads=n); kc.finalize(); torch.cuda.synchronize()
    kt = {k_: round(v[1], 2) for k_, v in kc.kernel_times().items()}
    print("context %d: l1 %.2f l2 %.2f count %.2f" % (r, kt.get("kc_l1_reads_kernel", 0), kt.get("kc_l2_split_kernel", 0) + kt.get("kc_l2_rec6_kernel", 0), kt.get("kc_count_kernel", 0)), flush=True)
    sys.stderr.flush()
    kc.close()
