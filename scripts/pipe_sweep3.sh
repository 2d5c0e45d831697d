#!/bin/bash
# chunk-shape sweep for the device-emit pipeline: LACX_PIPE_SPLIT weights
mkdir -p gpurun_out
rm -f gpurun_out/pipe_sweep3.txt
for sp in "5,5,4" "5,5,3,1" "6,5,3" "4,4,4,2" "5,4,3,2" "6,6,2" "1,1,1" "5,5,4,2" "6,5,4,3"; do
  v=$(LACX_PIPE_SPLIT=$sp python bench.py --no-cpu-baseline --steps 10 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "split $sp : $v" | tee -a gpurun_out/pipe_sweep3.txt
done
