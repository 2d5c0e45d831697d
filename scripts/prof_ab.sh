cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in batch3; do
  export LACX_LIB_OVERRIDE=$R/exp/liblacx_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$v -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads --no-decode-check > $R/gpurun_out/prof_$v.log 2>&1
  f=$(find $R/gpurun_out/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; cut -d, -f1-4 $f | sed 's/lacx:://; s/(.*)//' | head -14
done
