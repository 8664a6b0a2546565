# timing-only ablation of kc_count_kernel phases (results are wrong for any mask != 0)
for m in 0 64 128 192 1024; do echo "== KC_DEBUG_COUNT=$m"; KC_DEBUG_COUNT=$m python bench.py --steps 1 --warmup 1 --cpu-sample-reads 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['roofline']['kernels_ms'])"; done
