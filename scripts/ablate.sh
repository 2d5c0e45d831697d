for m in 0 1 2 4 6 8 16 32 64 127; do
  LACX_DEBUG_SKIP=$m timeout -k 10 120 python bench.py --seconds 120 --steps 2 --warmup 1 --no-cpu-baseline --analysis-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip=$m', d['breakdown_ms'])"
done
