#!/usr/bin/env python3
"""VGPR / SGPR counts and spill counts of every kernel in a host object with embedded gfx950 code
(llvm-readelf --notes of the device code object).  usage: scripts/kspill.py lossless-audio-codec_amd/build/k_analyze.o"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
obj = os.path.abspath(sys.argv[1])
with tempfile.TemporaryDirectory() as tmp:
    dev = os.path.join(tmp, "dev.o")
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={obj}",
                        "--targets=hip-amdgcn-amd-amdhsa--gfx950", f"--output={dev}"], capture_output=True)
    if r.returncode != 0 or not os.path.exists(dev) or os.path.getsize(dev) == 0:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={dev}"])
    notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", dev], text=True)
    size = subprocess.check_output([f"{LLVM}/llvm-readelf", "-S", dev], text=True)
keys = (".name", ".vgpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size")
rec = {}
for line in notes.splitlines():
    for k in keys:
        m = re.match(r"\s*-?\s*" + re.escape(k) + r":\s+(\S+)", line)
        if m:
            rec[k] = m.group(1)
    if ".wavefront_size" in line and ".vgpr_count" in rec:
        n = re.sub(r"^_ZN4lacx\d+", "", rec.get(".name", "?"))[:40]
        print(f"{n:42s} vgpr {rec.get('.vgpr_count')} sgpr {rec.get('.sgpr_count')} vspill {rec.get('.vgpr_spill_count')} "
              f"sspill {rec.get('.sgpr_spill_count')} scratch {rec.get('.private_segment_fixed_size')}")
        rec = {}
for line in size.splitlines():
    m = re.search(r"\.text\s+PROGBITS\s+\S+\s+\S+\s+(\S+)", line)
    if m:
        print("code bytes (.text):", int(m.group(1), 16))
