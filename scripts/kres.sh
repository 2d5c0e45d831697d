#!/bin/bash
# Kernel resource usage (VGPRs, SGPRs, scratch, LDS, occupancy) of the kernel units (k_front, k_analyze, k_emit, decode, wide) as hipcc reports it.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for u in k_front k_analyze k_emit decode wide; do
hipcc -O3 -std=c++20 --offload-arch=gfx950 $EXTRA -I$ROOT/lossless-audio-codec_amd/csrc -I$ROOT/include -c $ROOT/lossless-audio-codec_amd/csrc/$u.hip \
  -Rpass-analysis=kernel-resource-usage -o /tmp/kres.o 2>&1
done | python3 -c '
import sys, re
cur = None
for line in sys.stdin:
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        name = re.sub(r"^_ZN4lacx\d+", "", name)
        cur = name[:34]
        print()
        print(f"{cur:36s}", end="")
    elif any(t.startswith(k) for k in ("VGPRs:", "SGPRs:", "ScratchSize", "LDS Size", "Occupancy", "AGPRs")):
        print(t.replace(" [bytes/lane]", "").replace(" [bytes/block]", "").replace(" [waves/SIMD]", ""), end="  ")
print()
'
