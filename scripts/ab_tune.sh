# usage: ab_tune.sh [--reads N] tune1 tune2 ...   ("" = default geometry)
READS=50000000
if [ "$1" = "--reads" ]; then READS=$2; shift 2; fi
for t in "$@"; do echo "== $t"; python bench.py --reads $READS --steps 2 --warmup 1 --cpu-sample-reads 0 --check ${t:+--tune $t} 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']/1e9,2), d['roofline']['kernels_ms'], d['results']['total_kmers'], d['results']['num_unique'])
except Exception as e: print('failed', e)"; done
