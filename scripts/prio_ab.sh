for P in "0,1,-1" "-1,1,0" "-1,0,1" "-1,1,-1"; do
  LACX_STREAM_PRIO=$P python bench.py --no-cpu-baseline --no-decode-check > gpurun_out/r4u/b_$P.json 2> gpurun_out/r4u/b_$P.err
done
