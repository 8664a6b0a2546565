mkdir -p gpurun_out/r4
for k in 15 17 19 21; do
python3 bench.py --k $k --reads 30000000 --steps 3 --warmup 1 --cpu-sample-reads 0 --no-end-to-end 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print($k, round(d['ms_per_step'],2), round(d['value']/1e9,1), round(d['roofline']['frac'],3), d['roofline']['kernels_ms'])"
done
python3 bench.py --steps 3 --warmup 1 --cpu-sample-reads 0 --no-end-to-end 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('default', round(d['ms_per_step'],2), d['roofline']['kernels_ms'])"
