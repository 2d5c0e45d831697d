"""Diagnostic: per-phase shader cycles of k_analyze<16,1024> (needs a library built with EXTRA=-DLACX_STAMPS)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import numpy as np
pkg = ge.load_pkg()
import torch
secs = int(sys.argv[1]) if len(sys.argv) > 1 else 120
kind = sys.argv[2] if len(sys.argv) > 2 else "music"
L, R = pkg.synth.synth_pcm(secs * 48000, 2, 16, 48000, seed=2026, kind=kind)
dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
enc = pkg.lacx.Encoder(12, 2, 48000, 16, device=0)
lib = pkg.lacx.lib()
buf = (C.c_ulonglong * 32)()
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    enc.analyze_device(dl.data_ptr(), dr.data_ptr(), L.size, 0)
    lib.lacx_debug_stamps(buf)
names = ["stage", "score(prev)", "residual+bound", "store+scan1", "bound reduce", "B1 wait", "scan2+planes", "B2 wait", "phase_a", "-", "B3 wait", "-", "phase_b", "reduce", "B5 wait", "final score", "part: r+scan", "grp+scan", "seg_static", "part pass", "B wait", "choose+final"]
tot = sum(buf[i] for i in range(22))
waves = buf[24]
print(f"waves {waves}, cycles/wave {tot / max(1, waves):.0f}, full_ms {enc.timing().full_ms:.3f}, realtime ticks/wave {buf[22] / max(1, waves):.0f} -> shader clock {tot / max(1, buf[22]) * 0.1:.3f} GHz, wave lifetime {buf[22] / max(1, waves) / 100:.1f} us")
for i, nme in enumerate(names):
    print(f"  {nme:14s} {buf[i] / max(1, waves):10.0f} cyc/wave  {100.0 * buf[i] / max(1, tot):5.1f}%")
