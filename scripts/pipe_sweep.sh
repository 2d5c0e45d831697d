for ch in 1 2 4 8; do for th in 16 64 128 256; do
  LACX_PIPE_CHUNKS=$ch LACX_EMIT_THREADS=$th timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['breakdown_ms']; print('chunks=$ch threads=$th', d['value'], d['ms_per_step'], 'api', b['api_call'], 'tail', b['host_emit_tail'])"
done; done
