# bench.py with other chunk sizes of the two arenas (records per chunk): scripts/tune_chunks.sh "chunk1=2048" "chunk1=4096" ...
for t in "$@"; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample-reads 0 --no-end-to-end ${t:+--tune $t} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); k=d['roofline']['kernels_ms']
print('tune [$t]', round(d['ms_per_step'],2), 'l1', round(k.get('kc_l1_reads_kernel',0),2), 'l2', round(k.get('kc_l2_split_kernel',0)+k.get('kc_l2_rec6_kernel',0),2), 'count', round(k.get('kc_count_kernel',0),2), d['results']['total_kmers'], round(d['results']['table_GB'],1))"
done
