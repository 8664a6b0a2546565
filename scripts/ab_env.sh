# A/B of run-time switches of ONE library build on one box: ab_env.sh [-n ROUNDS] [-k K -r READS] "ENV1=.. ENV2=.." "ENV=.." ...
# (an empty string "" = the defaults).  Interleaved rounds, one process per run; prints min / median of step and kernels.
N=3; K=21; READS=50000000
while [ "$1" = "-n" ] || [ "$1" = "-k" ] || [ "$1" = "-r" ]; do
  case $1 in -n) N=$2;; -k) K=$2;; -r) READS=$2;; esac; shift 2
done
for r in $(seq $N); do for e in "$@"; do env $e timeout -k 10 300 python bench.py --k $K --reads $READS --steps 3 --warmup 1 --cpu-sample-reads 0 --no-end-to-end 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); k=d['roofline']['kernels_ms']
print('[$e]', round(d['ms_per_step'],2), round(k.get('kc_l1_reads_kernel',0)+k.get('kc_l1_reads16_kernel',0),2), round(k.get('kc_l2_split_kernel',0)+k.get('kc_l2_rec6_kernel',0),2), round(k.get('kc_count_kernel',0),2), d['results']['total_kmers'])" || exit 1; done; done | tee /tmp/abe.$$
python - <<PY
import collections,statistics
rows=collections.defaultdict(list)
for line in open('/tmp/abe.$$'):
    name,rest=line.rsplit(']',1)
    rows[name+']'].append([float(x) for x in rest.split()[:4]])
print('%-40s %s' % ('variant', 'min/median: step, l1, l2, count'))
for l,v in rows.items():
    cols=list(zip(*v))
    print('%-40s %s' % (l, '  '.join('%.1f/%.1f' % (min(c), statistics.median(c)) for c in cols)))
PY
