# N bench processes with the arena probe logging: does the probe's verdict match level 1's time?  probe_runs.sh [N] [lib]
N=${1:-6}
for r in $(seq $N); do
  KC_ARENA_PROBE=1 KC_LIB=$PWD/${2:-mhm2_kmer_analysis_v2_amd/csrc/libkcount_mi355.so} timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample-reads 0 --no-end-to-end 2>/tmp/pr.err | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); k=d['roofline']['kernels_ms']
print('run $r', round(d['ms_per_step'],2), 'l1', round(k.get('kc_l1_reads_kernel',0)+k.get('kc_l1_reads16_kernel',0),2), 'l2', round(k.get('kc_l2_split_kernel',0)+k.get('kc_l2_rec6_kernel',0),2), 'count', round(k.get('kc_count_kernel',0),2))"
  grep "arena probe" /tmp/pr.err
done
