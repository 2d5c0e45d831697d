#!/bin/bash
# Round profile of the bench command: kernel-trace stats (default pipeline and as one chunk), the SQ counter passes behind the
# VALU-side roofline, and the HBM traffic counters.  Every rocprofv3 run has the program directly after `--`, and PMC
# passes never share a run with a trace domain.
#   scripts/profile_valu.sh <tag> [extra bench.py args]      (on the GPU box; results under gpurun_out/<tag>/)
# then, in the build container:  python scripts/summarize_counters.py <tag>   -> profiles/<tag>_*.{csv,json}, profiles/valu.json
TAG=$1
shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
ARGS="--no-cpu-baseline --no-end-to-end --no-other-workloads --no-decode-check $@"
B="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $ARGS"
run() {  # name, env prefix vars..., then rocprof args
  local name=$1; shift
  echo "== $name" >> $OUT/progress.log
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 $B > $OUT/$name.log 2>&1 || echo "$name failed rc=$?" >> $OUT/progress.log
}
# Kernel traces: 10 timed steps behind 3 warm-up steps, so that the averages are not the cold first call's.
# trace: the bench command as it runs.  Under rocprofv3 the runtime moves device-to-host copies with blit KERNELS
# (the __amd_rocclr_copyBuffer rows) instead of the SDMA engines, so the copies that drain the payload beside the
# analysis take CUs from it there (k_analyze 2.5-2.7 ms in this trace against 2.05-2.1 ms unprofiled; HSA_ENABLE_SDMA=0
# without the profiler shows the same).  trace1: the same command with LACX_DIRECT_PACKER=1 (the packer writes the
# payload over PCIe itself, no copies): the kernel as undisturbed as it runs unprofiled.
B="$GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 $ARGS"
run trace --kernel-trace --stats
LACX_DIRECT_PACKER=1 run trace1 --kernel-trace --stats
B="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $ARGS"
export LACX_PIPE_CHUNKS=1
run sq1 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run grbm --pmc GRBM_GUI_ACTIVE
run ic1 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH
unset LACX_PIPE_CHUNKS
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
tail -3 $OUT/trace.log
cat $OUT/progress.log
