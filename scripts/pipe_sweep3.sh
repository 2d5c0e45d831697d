#!/bin/bash
# chunk-shape sweep for the device-emit pipeline: LACX_PIPE_SPLIT weights
mkdir -p gpurun_out
rm -f gpurun_out/pipe_sweep3.txt
for sp in "1,1,1" "3,3,2" "4,4,3" "2,2,2,1" "3,3,3,1" "1,1,1,1,1" "5,5,4" "6,5,4" "1,1,1,1,1,1"; do
  v=$(LACX_PIPE_SPLIT=$sp python bench.py --no-cpu-baseline --steps 10 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "split $sp : $v" | tee -a gpurun_out/pipe_sweep3.txt
done
