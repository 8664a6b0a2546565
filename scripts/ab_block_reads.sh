# the records flow in wire units at N = 1 by block size (bench.py --force-sharded --shard-flow records --block-reads B)
for B in 4000000 8000000 12500000 16700000 25000000; do
python3 bench.py --force-sharded --shard-flow records --steps 2 --warmup 1 --cpu-sample-reads 0 --no-end-to-end --block-reads $B 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('block-reads=$B', round(d['ms_per_step'],2), d['roofline']['kernels_ms'])"
done
