#!/bin/bash
# Tuning run: bench.py under a list of pipeline chunk splits (LACX_PIPE_SPLIT weights).  usage: sweep_split.sh <tag> split...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for sp in "$@"; do
  LACX_PIPE_SPLIT=$sp python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end > $OUT/split_$sp.json 2>/dev/null
  python - $OUT/split_$sp.json $sp <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"split {sys.argv[2]:10s} value {d['value']:9.1f}  ms/step {d['ms_per_step']:.3f}  k_full {d['breakdown_ms']['k_analyze_full']:.3f}  exec {d['roofline']['kernel_exec_ms']}x{d['roofline']['launches_per_step']}  front {d['breakdown_ms']['k_ingest_levinson']:.3f}+{d['breakdown_ms']['k_probe_decide']:.3f}  emit {d['breakdown_ms']['k_emit']:.3f}")
PY
done
