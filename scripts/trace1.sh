#!/bin/bash
# kernel trace of the bench as one pipeline chunk (no overlap between kernels): clean per-kernel durations
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
export LACX_PIPE_CHUNKS=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
for f in $(find $OUT/trace -name '*kernel_stats.csv'); do cut -c1-60,150-400 $f | sed 's/([^)]*)//g' | cut -c1-200; done
