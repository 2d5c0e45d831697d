#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of scripts/profile_round.sh:  update_traffic.py <tag> <launches_per_step>
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streaming reads on gfx950 (the staging
loads are global_load_dwordx4); WRITE_SIZE is taken as it is; both are KB per launch."""
import json, os, shutil, sys, glob

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, launches = sys.argv[1], int(sys.argv[2])
src = os.path.join(root, "gpurun_out", tag)
t = json.load(open(os.path.join(src, "traffic_counters.json")))
k = [x for x in t if "k_analyze" in x and "Geo<16, 1024" in x][0]
f, w = t[k]["FETCH_SIZE"]["mean"], t[k]["WRITE_SIZE"]["mean"]
blocks = 1758
out = {
    "kernel": "k_analyze<16,1024>",
    "command": "python bench.py --steps 3 --warmup 1 --no-cpu-baseline (default workload: interleaved int16 device PCM, "
               "device emit), rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (scripts/profile_round.sh)",
    "launches_per_step": launches,
    "fetch_size_kb_per_launch": f,
    "write_size_kb_per_launch": w,
    "traffic_bytes_per_launch": int(2 * f * 1024 + w * 1024),
    "algorithmic_bytes_per_launch": int((28_800_000 * 2 * 2 + 2 * blocks * 296) / launches),
    "note": "KB -> bytes x1024; FETCH_SIZE x2 (16-byte streaming reads on gfx950, MI355X_MICROARCH.md), WRITE_SIZE as is. "
            "Each PCM byte is read once: the two workgroups of a block share an XCD, so the second read is an L2 hit; "
            "the kernel has no spills.",
}
json.dump(out, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "traffic_counters.json"), os.path.join(root, "profiles", f"{tag}_traffic_counters_per_launch.json"))
for p in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(p, os.path.join(root, "profiles", f"{tag}_kernel_stats_bench.csv"))
print(out["traffic_bytes_per_launch"], out["algorithmic_bytes_per_launch"])
