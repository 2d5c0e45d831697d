#!/bin/bash
# Tuning run: bench.py per material kind.  usage: sweep_kinds.sh <tag> <bit_depth> <rate> kind...
TAG=$1; BD=$2; RATE=$3; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT
for k in "$@"; do
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --kind $k --bit-depth $BD --rate $RATE > $OUT/$k.json 2>$OUT/$k.err
  python - $OUT/$k.json $k <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    b = d['breakdown_ms']
    print(f"{sys.argv[2]:14s} value {d['value']:9.1f}  ms/step {d['ms_per_step']:7.3f}  k_full {b['k_analyze_full']:7.3f}  front {b['k_ingest_levinson']:.3f}+{b['k_probe_decide']:.3f}")
except Exception as ex:
    print(f"{sys.argv[2]:14s} FAILED {ex}")
PY
done
