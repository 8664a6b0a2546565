# the N > 1 code paths at N = 1 for two-word k-mers (BASELINE config 4's k): single pass and records flow, 30 M reads
for flow in single-pass records; do
python3 bench.py --k 51 --reads 30000000 --force-sharded --shard-flow $flow --steps 2 --warmup 1 --cpu-sample-reads 0 --no-end-to-end 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('k=51 $flow', round(d['ms_per_step'],2), round(d['value']/1e9,1), d['roofline']['kernels_ms'])"
done
python3 bench.py --k 51 --reads 30000000 --steps 2 --warmup 1 --cpu-sample-reads 0 --no-end-to-end 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('k=51 unsharded', round(d['ms_per_step'],2), round(d['value']/1e9,1), d['roofline']['kernels_ms'])"
