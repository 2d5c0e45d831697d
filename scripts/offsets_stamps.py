"""Diagnostic: where k_offsets spends its 10 or 50 microseconds (needs exp/liblacx_stamps.so from scripts/build_variant.sh
stamps -DLACX_STAMPS=1).  Per encode of the bench stream: 100 MHz realtime stamps of thread 0 -- start, loads back, scan
barrier passed, stores issued -- and the host-visible tail time.  usage: offsets_stamps.py [direct]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "direct":
    os.environ["LACX_DIRECT_PACKER"] = "1"
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_pkg()
lacx, synth = pkg.lacx, pkg.synth
lacx.use_library(os.path.join(ROOT, "exp", "liblacx_stamps.so"))
L, R = synth.synth_pcm(600 * 48000, 2, 16, 48000, seed=2026, kind="music", stereo="wide")
d = torch.from_numpy(synth.interleave(L, R, 16).view(np.int16)).cuda()
enc = lacx.Encoder(12, 2, 48000, 16, device=0)
buf = (C.c_ulonglong * 48)()
for it in range(14):
    enc.encode_shard_pcm_device_begin(d.data_ptr(), lacx.PCM_INTERLEAVED_I16, 2, L.size, 0)
    enc.encode_shard_end()
    lacx.lib().lacx_debug_stamps(buf)
    t = enc.timing()
    s = [buf[33 + i] for i in range(4)]
    print(f"call {it:2d}: loads {10 * (s[1] - s[0]):6d} ns, scan+barrier {10 * (s[2] - s[1]):6d} ns, stores {10 * (s[3] - s[2]):6d} ns; emit phase {t.emit_ms * 1e3:7.1f} us", flush=True)
