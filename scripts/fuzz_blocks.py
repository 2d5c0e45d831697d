"""Randomised parity sweep on the GPU (diagnostic, not part of the default suite): whole streams of random shape
and material through the device path against the oracle, plus single blocks through lacx_block_encode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as ge
import oracleshim

pkg = ge.load_pkg()
lacx, synth = pkg.lacx, pkg.synth
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 150
n_blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 600
KINDS = ["music", "noise", "silence", "near_silence", "sparse", "ramp", "walk", "tone", "mixed"]
STEREO = ["wide", "narrow", "identical", "anticorr", "independent"]


def punch_gaps(x):
    """Zero out random intervals (runs of every length class, many crossing the 16-sample chunk borders)."""
    if x is None or x.size < 8:
        return x
    x = x.copy()
    pos = 0
    while pos < x.size:
        keep = int(rng.choice([1, 1, 2, 3, 7, 16, 40, 300]))
        gap = int(rng.choice([1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 1000, 5000]))
        pos += keep
        x[pos:pos + gap] = 0
        pos += gap
    return x


bad = 0
t0 = time.time()
for it in range(n_streams):
    ch = int(rng.integers(1, 3))
    bd = int(rng.choice([16, 24]))
    sr = int(rng.choice([44100, 48000, 96000, 192000]))
    sm = int(rng.integers(0, 3)) if ch == 2 else 0
    frames = int(rng.choice([1, 2, 31, 33, 255, 4095, 4096, 4097, 16383, 16384, 16385, int(rng.integers(1, 70000))]))
    kind = str(rng.choice(KINDS))
    left, right = synth.synth_pcm(frames, ch, bd, sr, seed=int(rng.integers(1, 10**6)), kind=kind, stereo=str(rng.choice(STEREO)),
                                  start=int(rng.integers(0, 10**6)))
    if rng.random() < 0.3:  # scale down: low-level material exercises zero-run / bin modes
        sh = int(rng.integers(1, bd - 1))
        left = (left >> sh).astype(np.int32)
        right = None if right is None else (right >> sh).astype(np.int32)
    if rng.random() < 0.25:
        left, right = punch_gaps(left), punch_gaps(right)
    zr, pt = bool(rng.random() < 0.85), bool(rng.random() < 0.85)
    enc = lacx.Encoder(12, sm, sr, bd)
    enc.set_zero_run_enabled(zr)
    enc.set_partitioning_enabled(pt)
    enc.set_host_emit(bool(rng.random() < 0.3))
    got = enc.encode(left, right)
    want = oracleshim.encode(left, right, sr, bd, sm, zero_run=zr, partitioning=pt, threads=8)
    if got != want:
        bad += 1
        print("STREAM MISMATCH", it, dict(ch=ch, bd=bd, sr=sr, sm=sm, frames=frames, kind=kind, zr=zr, pt=pt))
    dl, dr, _, _ = lacx.decode(got)  # and back through the device decoder
    if not np.array_equal(dl, left) or (right is not None and not np.array_equal(dr, right)):
        bad += 1
        print("DECODE MISMATCH", it, dict(ch=ch, bd=bd, sr=sr, sm=sm, frames=frames, kind=kind, zr=zr, pt=pt))
print(f"streams: {n_streams - bad}/{n_streams} identical and decoded back, {time.time() - t0:.1f}s")
be = lacx.BlockEncoder()
badb = 0
t0 = time.time()
for it in range(n_blocks):
    n = int(rng.choice([1, 2, 3, 5, 13, 31, 32, 33, 63, 64, 65, 255, 256, 257, 1000, 4096, 8191, 16384, int(rng.integers(1, 16385))]))
    kind = str(rng.choice(KINDS))
    x, _ = synth.synth_pcm(n, 1, 24, 48000, seed=int(rng.integers(1, 10**6)), kind=kind, start=int(rng.integers(0, 10**6)))
    if rng.random() < 0.4:
        x = (x >> int(rng.integers(1, 23))).astype(np.int32)
    if rng.random() < 0.1:
        x = (x.astype(np.int64) * 2).clip(-(1 << 24), 1 << 24).astype(np.int32)  # side-channel range
    if rng.random() < 0.25:
        x = punch_gaps(x)
    got = be.encode(x)
    want = oracleshim.block_encode(x)
    if got != want:
        badb += 1
        print("BLOCK MISMATCH", it, n, kind)
print(f"blocks: {n_blocks - badb}/{n_blocks} identical, {time.time() - t0:.1f}s")
sys.exit(1 if (bad or badb) else 0)
