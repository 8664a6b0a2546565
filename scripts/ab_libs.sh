# A/B of library builds on ONE box: ab_libs.sh [-n ROUNDS] lib1.so lib2.so ...  (paths relative to the repo root).
# Interleaved rounds, one process per run (level 1 varies by +-10 % from process to process: compare minima and medians).
N=3
if [ "$1" = "-n" ]; then N=$2; shift 2; fi
for r in $(seq $N); do for l in "$@"; do KC_LIB=$PWD/$l timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample-reads 0 --no-end-to-end 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); k=d['roofline']['kernels_ms']
print('$l', round(d['ms_per_step'],2), round(k.get('kc_l1_reads_kernel',0)+k.get('kc_l1_reads16_kernel',0),2), round(k.get('kc_l2_split_kernel',0)+k.get('kc_l2_rec6_kernel',0),2), round(k.get('kc_count_kernel',0),2), d['results']['total_kmers'])" || exit 1; done; done | tee /tmp/ab.$$ 
python - <<PY
import collections,statistics
rows=collections.defaultdict(list)
for line in open('/tmp/ab.$$'):
    f=line.split()
    rows[f[0]].append([float(x) for x in f[1:5]])
print('%-30s %s' % ('lib', 'min/median: step, l1, l2, count'))
for l,v in rows.items():
    cols=list(zip(*v))
    print('%-30s %s' % (l.split('/')[-1], '  '.join('%.1f/%.1f' % (min(c), statistics.median(c)) for c in cols)))
PY
