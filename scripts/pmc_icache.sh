#!/bin/bash
# Diagnostic: instruction-cache counters of the bench command (the analysis kernel's code is several hundred KiB).
#   scripts/pmc_icache.sh <tag>    (on the GPU box; results under gpurun_out/<tag>/)
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
B="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-other-workloads --no-decode-check"
export LACX_PIPE_CHUNKS=1
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH --output-format csv -d $OUT/ic1 -- python3 $B > $OUT/ic1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_IFETCH_LEVEL SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/ic2 -- python3 $B > $OUT/ic2.log 2>&1

