#!/usr/bin/env python3
"""Prints the kernel timeline of the last bench step from a rocprofv3 --kernel-trace CSV (start/end relative to the step's
first kernel, stream/queue id).  usage: timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void lacx::", "").replace("lacx::", "").split("(")[0]
    return n.replace("Geo<16, 1024> ", "F").replace("Geo<4, 64> ", "P")[:26]
# steps are separated by gaps > 300 us
steps, cur, last_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > 300_000 and cur:
        steps.append(cur)
        cur = []
    cur.append(r)
    last_end = max(last_end or 0, e)
if cur:
    steps.append(cur)
st = steps[-1]
t0 = int(st[0]["Start_Timestamp"])
for r in st:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} -> {e / 1e3:9.1f} us  ({(e - s) / 1e3:8.1f})  q{r.get('Queue_Id', '?'):>3s}  {short(r['Kernel_Name'])}  grid {r.get('Grid_Size', '?')}")
print(f"step span {(max(int(r['End_Timestamp']) for r in st) - t0) / 1e3:.1f} us, {len(steps)} steps seen")
