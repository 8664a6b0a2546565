# A/B at another k: abk.sh K lib1 lib2
K=$1; shift
for r in 1 2 3; do for l in "$@"; do KC_LIB=$PWD/$l timeout -k 10 300 python bench.py --k $K --reads 30000000 --steps 3 --warmup 1 --cpu-sample-reads 0 --no-end-to-end 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); k=d['roofline']['kernels_ms']
print('$l', 'k=$K', round(d['ms_per_step'],2), round(k.get('kc_l1_reads_kernel',0),2), round(k.get('kc_l2_split_kernel',0)+k.get('kc_l2_rec6_kernel',0),2), round(k.get('kc_count_kernel',0),2), d['results']['total_kmers'])" || exit 1; done; done
