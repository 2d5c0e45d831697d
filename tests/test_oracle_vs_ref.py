"""Pins the oracle (oracle/lac_oracle.c) against the unmodified reference build (oracle/_ref)."""
import numpy as np
import pytest

KINDS = ["music", "noise", "silence", "near_silence", "sparse", "ramp", "walk", "tone", "mixed"]


@pytest.mark.parametrize("kind", KINDS)
def test_stream_bytes_equal_reference(pkg, oracle, ref, kind):
    for bd, sr, ch, sm in [(16, 48000, 2, 2), (24, 96000, 2, 2), (16, 44100, 1, 0), (24, 192000, 2, 1), (16, 48000, 2, 0)]:
        left, right = pkg.synth.synth_pcm(16384 * 2 + 777, ch, bd, sr, seed=5, kind=kind)
        assert oracle.encode(left, right, sr, bd, sm, threads=4) == ref.encode(left, right, sr, bd, sm)


def test_flags_and_small_sizes(pkg, oracle, ref):
    for n in (1, 2, 12, 13, 31, 32, 33, 255, 256, 257, 4095, 4096, 4097):
        left, right = pkg.synth.synth_pcm(n, 2, 16, 48000, seed=n, kind="noise", stereo="independent")
        assert oracle.encode(left, right, 48000, 16, 2) == ref.encode(left, right, 48000, 16, 2)
    left, right = pkg.synth.synth_pcm(16384 + 99, 2, 24, 48000, seed=3, kind="mixed")
    for zr in (False, True):
        for pt in (False, True):
            assert oracle.encode(left, right, 48000, 24, 2, zr, pt) == ref.encode(left, right, 48000, 24, 2, zr, pt)
            assert oracle.block_encode(left[:5000], zr, pt) == ref.block_encode(left[:5000], zr, pt)


def test_lpc_and_adapt_primitives(pkg, oracle, ref):
    left, _ = pkg.synth.synth_pcm(2048, 1, 24, 48000, seed=9, kind="music")
    for order in (4, 6, 8, 10, 12):
        u1, c1 = oracle.lpc_analyze(left, order)
        u2, c2 = ref.lpc_analyze(left, order)
        assert u1 == u2 and np.array_equal(c1, c2)
    rng = np.random.default_rng(1)
    for scale in (3, 300, 70000, 1 << 22):
        u = rng.integers(0, scale, size=3000, dtype=np.uint32)
        u[500:900] = 0
        assert np.array_equal(oracle.adapt_k_sequence(u), ref.adapt_k_sequence(u))


def test_decoder_roundtrip_and_matches_reference_decoder(pkg, oracle, ref):
    left, right = pkg.synth.synth_pcm(16384 * 2 + 5, 2, 24, 96000, seed=4, kind="mixed")
    data = ref.encode(left, right, 96000, 24, 2)
    l1, r1, h1 = oracle.decode(data)
    l2, r2, h2 = ref.decode(data)
    assert np.array_equal(l1, left) and np.array_equal(r1, right)
    assert np.array_equal(l1, l2) and np.array_equal(r1, r2) and h1 == h2


def test_wide_blocks_oracle_matches_reference(oracle, ref):
    """Block::Encoder outside the 25-bit domain of validated input: residuals that leave int32 and the order fallback
    (ref lpc.cpp:24-36, 188-229), 32-bit zigzag values, k = 31.  Pins the oracle where the GPU's wide kernel is checked
    against it (tests/test_gpu_parity.py::test_block_encoder_full_int32_domain)."""
    rng = np.random.default_rng(77)
    n = 4096
    t = np.arange(n)
    cases = []
    for bits in (26, 28, 30, 31):
        amp = (1 << (bits - 1)) - 1
        cases.append((np.sin(t / 37.0) * amp * 0.9).astype(np.int64))
        cases.append(rng.integers(-amp, amp, size=n))
        cases.append(np.where((t // 64) % 2 == 0, amp, -amp).astype(np.int64) + rng.integers(-1000, 1000, size=n))
    alt = np.where(t % 2 == 0, 2**31 - 1, -2**31).astype(np.int64)
    cases += [alt, np.concatenate([alt[:2000], np.zeros(2096, np.int64)]), rng.integers(-2**30, 2**30, size=37)]
    for i, x in enumerate(cases):
        x = np.clip(x, -2**31, 2**31 - 1).astype(np.int32)
        for zr, pt in ((True, True), (False, False)):
            assert oracle.block_encode(x, zr, pt) == ref.block_encode(x, zr, pt), (i, zr, pt)
