for r in 1 2 3 4 5 6 7 8; do
  KC_ARENA_PROBE=0 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample-reads 0 --no-end-to-end 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); k=d['roofline']['kernels_ms']
print('run $r (no probe, no scrub)', round(d['ms_per_step'],2), 'l1', round(k.get('kc_l1_reads_kernel',0)+k.get('kc_l1_reads16_kernel',0),2), 'l2', round(k.get('kc_l2_split_kernel',0)+k.get('kc_l2_rec6_kernel',0),2), 'count', round(k.get('kc_count_kernel',0),2))"
done
