"""The two tiny-chunks geometries that ended with KC_ERR_CAPACITY in round 1 (gpurun_out/stress2.log cases 4 and 8), at that scale."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import mhm2_kmer_analysis_v2_amd as pkg
from oracle import cpu_oracle as O
from helpers import random_reads
cases = [(13, 116640, 840240, dict(writers=3, p1=64, p2=4, slots=2048, chunk1=16, chunk2=16, chain1_max=36, chain2_max=17, ovf_capacity=1048576)),
         (23, 153270, 791250, dict(writers=1, p1=64, p2=4, slots=2048, chunk1=16, chunk2=16, chain1_max=12, chain2_max=23, ovf_capacity=1048576))]
bad = 0
for i, (k, nreads, genome, t) in enumerate(cases):
    rng = np.random.default_rng(100 + i)
    reads, quals = random_reads(rng, nreads, min_len=k, max_len=k + 120, genome_len=genome, err=0.005, n_rate=0.01)
    b, q, offs = O.reads_to_arrays(reads, quals)
    o = O.Oracle(k, nranks=4, nthreads=8); o.add_reads(b, q, offs); want = o.finalize(); o.close()
    with pkg.KmerCounter(k, max_kmers_buffered=(1 << 22) * 30, tuning=t) as kc:
        kc.submit_reads(b, q, offs)
        got = kc.sorted_results(); st = kc.stats()
    ok = all(g.shape == w.shape and (g == w).all() for g, w in zip(got, want))
    print("case", i, "k", k, "ok" if ok else "MISMATCH", len(want[1]), "k-mers, table capacity", st["capacity"], flush=True)
    bad += not ok
sys.exit(1 if bad else 0)
