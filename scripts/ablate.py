"""Where does a kernel's time go?  Experiment builds only (scripts/fastbuild.sh OUT.so -DKC_ABLATE): one normal step
fills the arenas, then the same step again with one part of one kernel left out (its results are wrong, the data the later
kernels read is the first step's): KC_LIB=build/fast/abl.so python scripts/ablate.py [reads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mhm2_kmer_analysis_v2_amd as pkg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
k, L = 21, 150
dev = torch.device("cuda", 0)
b = torch.empty(n * L, dtype=torch.uint8, device=dev); q = torch.empty_like(b); o = torch.empty(n + 1, dtype=torch.int64, device=dev)
est = int(64 * 4_000_000 + n * L * 0.005 * k * 1.05) + (1 << 20)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
def context():
    kc = pkg.KmerCounter(k, device=0, max_elems=est, time_kernels=True, max_kmers_buffered=int(n * (L - k - 1) * 1.02) + (1 << 20))
    kc.set_stream(s.cuda_stream)
    return kc
kc = context()
kc.synth_reads_device(b, q, o, n, first_read=0, params=pkg.synth_params())
names = {"KC_ABL_L1": "kc_l1_reads16_kernel", "KC_ABL_L2": "kc_l2_rec6_kernel", "KC_ABL_COUNT": "kc_count_kernel"}
what = {1: "no global stores / no counter adds", 2: "no copy-out / no probes", 3: "no scatter, no copy-out / no vote + write", 4: "runs forced to whole aligned 64-byte blocks"}
def step(kc, finalize=True):
    kc.reset(); kc.kernel_times(clear=True)
    kc.submit_reads(b, q, o, nreads=n); kc.flush()
    if finalize:
        kc.finalize()
    torch.cuda.synchronize()
    return {k_: round(v[1], 2) for k_, v in kc.kernel_times().items()}
for it in range(2):
    print("full", step(kc), flush=True)
kc.close()
# a fresh context per experiment (zeroed arenas: what an ablated kernel does not write reads as k-mer 0 further down);
# level 1's are not finalized (level 2 cannot digest a bucket of identical records)
for var, kern in names.items():
    for a in ((1, 2, 3, 4) if var != "KC_ABL_COUNT" else (1, 2, 3)):
        os.environ[var] = str(a)
        kc = context()
        t = [step(kc, finalize=var != "KC_ABL_L1").get(kern, 0) for _ in range(2)]
        print("%s=%d (%s): %s %s ms" % (var, a, what[a], kern, t), flush=True)
        kc.close()
    os.environ[var] = "0"
os._exit(0)
