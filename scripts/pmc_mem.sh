#!/bin/bash
# vector-memory latency / TLB counters of the kernels (own PMC passes, no trace domains); $2.. = extra program
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
export LACX_PIPE_CHUNKS=1
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'lacx' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:36s} n={len(v):3d} mean={sum(v)/len(v):.4g}')
PY
