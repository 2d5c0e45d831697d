for ch in 1 2 3 4 6 8 12 16; do
  LACX_PIPE_CHUNKS=$ch timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['breakdown_ms']; print('chunks=$ch', d['value'], d['ms_per_step'], 'api', b['api_call'], 'full/launch', d['roofline']['kernel_ms'])"
done
