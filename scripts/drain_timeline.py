"""Diagnostic: the copy-engine drain of one device-resident encode as the library sees it (LACX_DEBUG_DRAIN=1: when
every progress range was announced, when the kernels were done, the call's total) -- stderr, four calls of the 10 min stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LACX_DEBUG_DRAIN"] = "1"
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_pkg(); lacx, synth = pkg.lacx, pkg.synth
frames = 28_800_000
L, R = synth.synth_pcm(frames, 2, 16, 48000, seed=2026, kind="music", stereo="wide")
inter = synth.interleave(L, R, 16)
d = torch.from_numpy(inter.view(np.int16)).cuda()
enc = lacx.Encoder(12, 2, 48000, 16, device=0)
for i in range(4):
    sys.stderr.write(f"--- call {i}\n"); sys.stderr.flush()
    enc.encode_shard_pcm_device_begin(d.data_ptr(), lacx.PCM_INTERLEAVED_I16, 2, frames, 0)
    enc.encode_shard_end()
    t = enc.timing()
    sys.stderr.write(f"total {t.total_ms:.3f} analysis {t.analysis_ms:.3f} kernels_done {t.kernels_done_ms:.3f}\n")
