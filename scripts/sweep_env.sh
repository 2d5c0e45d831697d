#!/bin/bash
# Tuning run: bench.py under a list of environment settings ("A=1 B=2" strings).  usage: sweep_env.sh <tag> "ENV..." ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end $BENCH_ARGS > $OUT/run_$i.json 2>$OUT/run_$i.err
  python - $OUT/run_$i.json "$envs" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"{sys.argv[2]:60s} value {d['value']:9.1f}  ms/step {d['ms_per_step']:.3f}  k_full {d['breakdown_ms']['k_analyze_full']:.3f}  exec {d['roofline']['kernel_exec_ms']}x{d['roofline']['launches_per_step']}  front {d['breakdown_ms']['k_ingest_levinson']:.3f}+{d['breakdown_ms']['k_probe_decide']:.3f}  emit {d['breakdown_ms']['k_emit']:.3f}  fused {d.get('fused_emit')}")
except Exception as ex:
    print(f"{sys.argv[2]:60s} FAILED {ex}")
PY
done
