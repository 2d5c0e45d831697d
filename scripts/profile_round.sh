#!/bin/bash
# Round profile: kernel trace stats + HBM traffic counters of the default bench.py command.
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
python3 - $OUT <<'PY'
import sys, glob, csv, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, d in agg.items():
    res[k] = {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in d.items()}
json.dump(res, open(out + '/traffic_counters.json', 'w'), indent=1)
for k, d in res.items():
    print(k, {c: round(x["mean"], 1) for c, x in d.items()})
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    print(open(f).read())
PY
