"""Integer-only, platform-independent synthetic PCM for parity tests and bench.py.

Every value is produced with numpy uint64/int64 arithmetic (no libm), so the GPU box regenerates
bit-identical inputs for the golden digests minted in the build container.
Shapes follow the reference's own test generators (restated, not copied):
  LCG/white noise        /root/reference/tests/test_lpc.cpp:101-109
  ramp / near-silence    /root/reference/tests/test_lpc.cpp:111-135
  sparse +-1, spikes     /root/reference/tests/test_zerorun.cpp:30-51,509-546
  stereo families        /root/reference/tests/test_e2e.cpp:666-810
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_MIX1 = np.uint64(0xBF58476D1CE4E5B9)
_MIX2 = np.uint64(0x94D049BB133111EB)

KINDS = ("music", "noise", "silence", "near_silence", "sparse", "ramp", "walk", "tone", "mixed")
STEREO = ("wide", "identical", "anticorr", "independent", "half_silent", "narrow")


def _hash64(n: np.ndarray, seed: int) -> np.ndarray:
    """splitmix64 finaliser of (n + seed*GOLD); counter based so any range can be generated."""
    with np.errstate(over="ignore"):
        z = n.astype(np.uint64) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF) * _GOLD
        z = (z + _GOLD) & _M64
        z ^= z >> np.uint64(30)
        z = z * _MIX1
        z ^= z >> np.uint64(27)
        z = z * _MIX2
        z ^= z >> np.uint64(31)
    return z


def _isin(phase32: np.ndarray, amp: int) -> np.ndarray:
    """Integer sine (Bhaskara I rational form): phase32 in [0,2^32) -> int64 in [-amp, amp]."""
    p = phase32.astype(np.int64)
    neg = p >= (1 << 31)
    p = np.where(neg, p - (1 << 31), p)  # half wave, [0, 2^31)
    q = (p * ((1 << 31) - p)) >> 31  # x(1-x) scaled by 2^31, <= 2^29
    num = (16 * q) * int(amp)
    den = 5 * (1 << 31) - 4 * q
    v = num // den
    return np.where(neg, -v, v)


def _tri(phase32: np.ndarray, amp: int) -> np.ndarray:
    """Triangle envelope in [0, amp]."""
    p = phase32.astype(np.int64)
    p = np.where(p >= (1 << 31), (1 << 32) - p, p)  # [0, 2^31]
    return (p * int(amp)) >> 31


def _phase(n: np.ndarray, step: int, phase0: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        return (n.astype(np.uint64) * np.uint64(step) + np.uint64(phase0)) & np.uint64(0xFFFFFFFF)


def _step_for(freq_mhz: int, sample_rate: int) -> int:
    """Phase increment for a frequency given in milli-hertz (integer arithmetic only)."""
    return ((freq_mhz << 32) // (1000 * sample_rate)) & 0xFFFFFFFF


def _signed_noise(n: np.ndarray, seed: int, bits: int) -> np.ndarray:
    """Uniform signed integers in [-2^(bits-1), 2^(bits-1))."""
    if bits <= 0:
        return np.zeros(n.shape, dtype=np.int64)
    h = _hash64(n, seed) >> np.uint64(64 - bits)
    return h.astype(np.int64) - (1 << (bits - 1))


def _voice(n, sample_rate, full, seed, base_mhz):
    """A handful of partials with slow, independent envelopes plus low-level noise."""
    out = np.zeros(n.shape, dtype=np.int64)
    ratios = (1000, 2003, 2996, 5502, 9017, 14011, 23003)  # per-mille multiples of the base
    amps = (300, 170, 110, 80, 60, 40, 30)  # per-mille of full scale (sum 0.79)
    for i, (ratio, a) in enumerate(zip(ratios, amps)):
        step = _step_for(base_mhz * ratio // 1000, sample_rate)
        ph0 = int(_hash64(np.array([i], dtype=np.uint64), seed + 17)[0] & np.uint64(0xFFFFFFFF))
        estep = 2000 + 977 * i + (seed % 13) * 101  # envelope periods of a few seconds
        env = _tri(_phase(n, estep, ph0 >> 1), 1 << 16)
        s = _isin(_phase(n, step, ph0), full * a // 1000)
        out += (s * env) >> 16
    return out


def _gen_mono(kind, n, sample_rate, bit_depth, seed):
    full = (1 << (bit_depth - 1)) - 1
    if kind == "silence":
        return np.zeros(n.shape, dtype=np.int64)
    if kind == "near_silence":
        return _signed_noise(n, seed, 2)  # [-2, 1]
    if kind == "noise":
        return _signed_noise(n, seed, bit_depth)
    if kind == "sparse":
        h = _hash64(n, seed)
        hit = (h & np.uint64(0x3F)) == 0  # 1 in 64 non-zero
        big = ((h >> np.uint64(8)) & np.uint64(0xFF)) == 0  # rare full-scale spikes
        sign = np.where(((h >> np.uint64(20)) & np.uint64(1)) == 1, -1, 1).astype(np.int64)
        v = np.where(hit, sign, 0).astype(np.int64)
        return np.where(hit & big, sign * (full >> 1), v)
    if kind == "ramp":
        period = 4096
        r = (n.astype(np.int64) % period) - period // 2
        return (r * (full // period)).astype(np.int64)
    if kind == "tone":
        return _isin(_phase(n, _step_for(440_000, sample_rate), seed & 0xFFFFFFFF), full * 9 // 10)
    if kind == "walk":
        # bounded pseudo random walk: triangle-folded sum of two incommensurate slow phases + steps
        a = _tri(_phase(n, 40503 + (seed % 7) * 11, seed * 7919), full) - full // 2
        b = _signed_noise(n >> np.uint64(3), seed + 5, max(2, bit_depth - 6))
        return np.clip(a + b, -full - 1, full)
    if kind == "music":
        v = _voice(n, sample_rate, full, seed, 196_000 + (seed % 5) * 27_500)
        # noise floor whose level breathes slowly (changes the best Rice parameter per partition)
        nenv = _tri(_phase(n, 9001 + (seed % 11) * 313, seed * 104729), 1 << 16)
        nb = max(2, bit_depth - 6)
        one = np.uint64(1)
        shaped = (_signed_noise(n, seed + 99, nb) + _signed_noise(n - one, seed + 99, nb)
                  + (_signed_noise(n - one - one, seed + 99, nb) >> 1)
                  - (_signed_noise(n - one - one - one, seed + 99, nb) >> 2))  # coloured floor
        v = v + ((shaped * nenv) >> 16)
        return np.clip(v, -full - 1, full)
    raise ValueError(f"unknown kind {kind!r}")


_MIXED_PLAN = ("music", "silence", "music", "near_silence", "noise", "sparse", "music", "ramp", "walk", "tone")


def synth_pcm(frames: int, channels: int = 2, bit_depth: int = 16, sample_rate: int = 48000,
              seed: int = 1, kind: str = "music", stereo: str = "wide", start: int = 0,
              total_frames: int | None = None, chunk: int = 1 << 20):
    """Returns (left, right) int32 arrays for frames [start, start+frames) of the named stream.

    `right` is None for mono. `kind="mixed"` switches material every `sample_rate*3` frames following
    a fixed plan so that zero-run, bin and static-Rice residual modes and partition orders > 0 are
    all exercised (BASELINE config 3). Values are always inside the bit-depth range.
    """
    assert bit_depth in (16, 24) and channels in (1, 2)
    left = np.empty(frames, dtype=np.int32)
    right = np.empty(frames, dtype=np.int32) if channels == 2 else None
    lo, hi = -(1 << (bit_depth - 1)), (1 << (bit_depth - 1)) - 1
    pos = 0
    while pos < frames:
        m = min(chunk, frames - pos)
        n = np.arange(start + pos, start + pos + m, dtype=np.uint64)
        if kind == "mixed":
            seg_len = sample_rate * 3
            seg = (n // np.uint64(seg_len)).astype(np.int64)
            a = np.zeros(m, dtype=np.int64)
            b = np.zeros(m, dtype=np.int64)
            for s in np.unique(seg):
                sel = seg == s
                k = _MIXED_PLAN[int(s) % len(_MIXED_PLAN)]
                st = STEREO[int(s) % len(STEREO)]
                aa, bb = _gen_stereo(k, st, n[sel], sample_rate, bit_depth, seed + int(s))
                a[sel] = aa
                b[sel] = bb
        else:
            a, b = _gen_stereo(kind, stereo, n, sample_rate, bit_depth, seed)
        left[pos:pos + m] = np.clip(a, lo, hi).astype(np.int32)
        if right is not None:
            right[pos:pos + m] = np.clip(b, lo, hi).astype(np.int32)
        pos += m
    return left, right


def _gen_stereo(kind, stereo, n, sample_rate, bit_depth, seed):
    a = _gen_mono(kind, n, sample_rate, bit_depth, seed)
    if stereo == "identical":
        return a, a.copy()
    if stereo == "anticorr":
        return a, -a
    if stereo == "half_silent":
        return a, np.zeros_like(a)
    b = _gen_mono(kind, n, sample_rate, bit_depth, seed + 1000003)
    if stereo == "independent":
        return a, b
    if stereo == "narrow":  # strongly correlated: side channel is small
        return a + (b >> 6), a - (b >> 6)
    # "wide": slowly panning mix of two voices, so LR and MS each win on some blocks
    pan = _tri(_phase(n, 15013 + (seed % 17) * 211, seed * 31337), 1 << 12)  # [0, 4096]
    return (a * (8192 - pan) + b * pan) >> 13, (a * pan + b * (8192 - pan)) >> 13


def interleave(left: np.ndarray, right: np.ndarray | None, bit_depth: int) -> np.ndarray:
    """WAV data-chunk layout: interleaved little-endian int16, or packed 3-byte int24 (as uint8)."""
    chans = [left] if right is None else [left, right]
    inter = np.stack(chans, axis=1).reshape(-1)
    if bit_depth == 16:
        return inter.astype("<i2")
    u = inter.astype(np.int32).view(np.uint32)
    out = np.empty((inter.size, 3), dtype=np.uint8)
    out[:, 0] = u & 0xFF
    out[:, 1] = (u >> 8) & 0xFF
    out[:, 2] = (u >> 16) & 0xFF
    return out.reshape(-1)
