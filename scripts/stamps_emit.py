"""Diagnostic: per-phase shader cycles of k_emit<16,1024> (needs a library built with EXTRA=-DLACX_STAMPS=2)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import numpy as np
pkg = ge.load_pkg()
import torch
secs = int(sys.argv[1]) if len(sys.argv) > 1 else 120
kind = sys.argv[2] if len(sys.argv) > 2 else "music"
L, R = pkg.synth.synth_pcm(secs * 48000, 2, 16, 48000, seed=2026, kind=kind)
inter = pkg.synth.interleave(L, R, 16)
d = torch.from_numpy(inter.view(np.int16)).cuda()
enc = pkg.lacx.Encoder(12, 2, 48000, 16, device=0)
lib = pkg.lacx.lib()
buf = (C.c_ulonglong * 32)()
for it in range(3):
    enc.encode_shard_pcm_device_view(d.data_ptr(), pkg.lacx.PCM_INTERLEAVED_I16, 2, L.size, 0)
    lib.lacx_debug_stamps(buf)
names = ["stage", "plan+B", "phase_r+nz", "scans pz/nx", "phase_a+scanF", "walk1", "B wait", "bit scan", "tile clear+B",
         "walk2", "B wait", "copy-out", "B wait"]
tot = sum(buf[i] for i in range(22))
waves = buf[24]
print(f"waves {waves}, cycles/wave {tot / max(1, waves):.0f}, emit_ms {enc.timing().emit_ms:.3f}, wave lifetime {buf[22] / max(1, waves) / 100:.1f} us")
for i, nme in enumerate(names):
    print(f"  {nme:14s} {buf[i] / max(1, waves):10.0f} cyc/wave  {100.0 * buf[i] / max(1, tot):5.1f}%")
