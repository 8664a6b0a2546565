for m in "$@"; do echo "== KC_DEBUG_COUNT=$m"; KC_DEBUG_COUNT=$m python bench.py --steps 1 --warmup 1 --cpu-sample-reads 0 2>&1 | grep -E "count kernel cycles" | head -1; done
