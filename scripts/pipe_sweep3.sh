#!/bin/bash
# chunk-shape sweep for the device-emit pipeline: LACX_PIPE_SPLIT weights
mkdir -p gpurun_out
for sp in "1" "1,1" "3,2" "2,1" "3,1" "4,2,1" "3,2,1" "5,3,1" "4,3,2,1" "8,4,2,1" "6,4,2,1,1" "2,2,1" "3,3,1,1"; do
  v=$(LACX_PIPE_SPLIT=$sp python bench.py --no-cpu-baseline --steps 10 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "split $sp : $v" | tee -a gpurun_out/pipe_sweep3.txt
done
