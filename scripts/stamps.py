"""Diagnostic: per-phase shader cycles of k_analyze<16,1024> (needs a library built with EXTRA=-DLACX_STAMPS).
usage: stamps.py [seconds] [kind] [iterations] [bit_depth]   -- runs the device-emit encode (fused emit included)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import numpy as np
pkg = ge.load_pkg()
import torch
secs = int(sys.argv[1]) if len(sys.argv) > 1 else 120
kind = sys.argv[2] if len(sys.argv) > 2 else "music"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
bd = int(sys.argv[4]) if len(sys.argv) > 4 else 16
L, R = pkg.synth.synth_pcm(secs * 48000, 2, bd, 48000, seed=2026, kind=kind)
inter = pkg.synth.interleave(L, R, bd)
d = torch.from_numpy(inter.view(np.int16) if bd == 16 else inter).cuda()
enc = pkg.lacx.Encoder(12, 2, 48000, bd, device=0)
lib = pkg.lacx.lib()
buf = (C.c_ulonglong * 40)()
layout = pkg.lacx.PCM_INTERLEAVED_I16 if bd == 16 else pkg.lacx.PCM_INTERLEAVED_I24
for it in range(iters):
    enc.encode_shard_pcm_device_view(d.data_ptr(), layout, 2, L.size, 0)
    lib.lacx_debug_stamps(buf)
names = ["stage", "score+select", "pass1: all bounds", "resid+store+scan1", "Bsel wait", "pass1 barrier", "scan2+planes", "-", "phase_a", "-", "B3 wait", "-", "phase_b", "reduce", "B5 wait", "-", "part: r+scan", "grp+scan", "seg_static", "part pass", "B wait", "choose+final", "(realtime)", "emit: plan+nx scan", "emit: phase A", "emit: walk 1", "emit: bit scan", "emit: zero tile", "emit: walk 2", "emit: barrier", "emit: copy+publish", "-"]
idx = [i for i in range(32) if i != 22]
tot = sum(buf[i] for i in idx)
waves = buf[32]
t = enc.timing()
print(f"waves {waves}, cycles/wave {tot / max(1, waves):.0f}, full_ms {t.full_ms:.3f}, realtime ticks/wave {buf[22] / max(1, waves):.0f} -> shader clock {tot / max(1, buf[22]) * 0.1:.3f} GHz, wave lifetime {buf[22] / max(1, waves) / 100:.1f} us")
for i in idx:
    print(f"  {names[i]:14s} {buf[i] / max(1, waves):10.0f} cyc/wave  {100.0 * buf[i] / max(1, tot):5.1f}%")
