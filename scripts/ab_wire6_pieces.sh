mkdir -p gpurun_out/r4
for q in 0 1 2 3 4; do
KC_WIRE6_LG_PIECES=$q python3 bench.py --force-sharded --shard-flow records --steps 2 --warmup 1 --cpu-sample-reads 0 --no-end-to-end --block-reads 50000000 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('lgQ=$q one block', round(d['ms_per_step'],2), d['roofline']['kernels_ms'])"
done
