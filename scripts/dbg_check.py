import os, sys
sys.path.insert(0, os.getcwd())
import mhm2_kmer_analysis_v2_amd as pkg
p = pkg.synth_params(num_genomes=4, min_genome_len=20000, max_genome_len=30000)
b, q, o = pkg.synth_reads_host(20000, params=p)
for v in ("0", "64", "64t", "128", "2048", "4096", "65536"):
    os.environ["KC_DEBUG_COUNT"] = v.rstrip("t")
    with pkg.KmerCounter(21, time_kernels=True, tuning=dict(mode=1) if v.endswith("t") else None) as kc:
        kc.submit_reads(b, q, o)
        kc.flush()
        st0 = kc.stats()
        res = kc.results()
        st = kc.stats()
        print(v, len(res[1]), st0["kmers_inserted"], st["num_unique"], st["sum_counts"], st["num_purged"], kc.kernel_times())
