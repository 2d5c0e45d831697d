#!/bin/bash
# Tuning run: the host-to-host leg (end_to_end) of bench.py under different upload pipelines.
# usage: sweep_e2e.sh <tag> "ENV=VAL ..." ...     (each argument: environment assignments for one run, "" = defaults)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$i.json 2>$OUT/$i.err
  python - $OUT/$i.json "$cfg" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    e = d['end_to_end']
    print(f"{sys.argv[2]:40s} e2e {e['ms']:7.3f} ms {e['value']:9.1f} Ms/s  identical {e['byte_identical']}   (resident {d['ms_per_step']:.3f} ms)")
except Exception as ex:
    print(f"{sys.argv[2]:40s} FAILED {ex}")
PY
done
