"""Measurement (not part of the default suite): the device decoder on the bench stream.  Encodes `seconds` of the synthetic
stereo 16/48 stream on the GPU, decodes the .lac `iters` times with lacx_decode, checks the PCM, prints kernel milliseconds.
usage: decode_bench.py [seconds] [iters] [kind]      (under rocprofv3 --kernel-trace --stats for profiles/)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_pkg()
lacx, synth = pkg.lacx, pkg.synth
secs = int(sys.argv[1]) if len(sys.argv) > 1 else 600
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = sys.argv[3] if len(sys.argv) > 3 else "music"
sr, bd = 48000, 16
left, right = synth.synth_pcm(secs * sr, 2, bd, sr, seed=2026, kind=kind, stereo="wide")
lac = lacx.Encoder(12, 2, sr, bd, device=0).encode(left, right)
best = 1e9
for _ in range(iters):
    t0 = time.perf_counter()
    dl, dr, info, ms = lacx.decode(lac)
    wall = (time.perf_counter() - t0) * 1e3
    best = min(best, ms)
    assert np.array_equal(dl, left) and np.array_equal(dr, right)
print(f"{kind}: {secs} s stereo {bd}/{sr // 1000}: {info.blocks} blocks, {len(lac)} B .lac; decode kernels {best:.2f} ms = "
      f"{2 * secs * sr / best / 1e3:.0f} Msamples/s (wall incl. H2D of the .lac and D2H of the PCM {wall:.0f} ms); PCM identical")
