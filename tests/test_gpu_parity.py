"""GPU parity tests: the HIP path (through the C ABI) against the oracle on the same seeded inputs.

Bit-exact is the bar everywhere: bytes of the .lac, plan records, autocorrelation, Q15 coefficients.
"""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu(pkg):
    if pkg.lacx.device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests need an MI355X (the product has no CPU fallback)")
    return pkg


def _ms(l, r):
    m = ((l.astype(np.int64) + r) >> 1).astype(np.int32)
    s = (l - r).astype(np.int32)
    return m, s


KINDS = ["music", "noise", "silence", "near_silence", "sparse", "ramp", "walk", "tone", "mixed"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("bit_depth", [16, 24])
def test_lpc_kernels_match_oracle(gpu, oracle, kind, bit_depth):
    """k_ingest autocorrelation (exact int64) and k_levinson (software x87) vs the oracle's long double."""
    left, right = gpu.synth.synth_pcm(16384 + 300, 2, bit_depth, 48000, seed=21, kind=kind)
    m, s = _ms(left, right)
    be = gpu.lacx.BlockEncoder()
    for x in (left[:16384], s[:16384], m[16384:], left[100:100 + 256], s[5:18]):
        ac, coef, used = be.debug_lpc(x)
        assert np.array_equal(ac, oracle.autocorr(x, 12))
        mvo = min(32, x.size - 1)
        for ci, cand in enumerate((4, 6, 8, 10, 12)):
            if cand > mvo:
                assert used[ci] == 0
                continue
            ou, oc = oracle.lpc_analyze(x, cand)
            assert used[ci] == ou
            assert np.array_equal(coef[ci][1:cand + 1], oc[1:cand + 1])


def _check_plan(pl, op, nbytes):
    assert pl.valid == 1
    assert (pl.predictor_type, pl.order, pl.partition_order, pl.total_bits) == \
        (op.predictor_type, op.order, op.partition_order, op.total_bits)
    if op.predictor_type == 2:
        assert [pl.coef[i] for i in range(op.order)] == [op.coeffs_q15[i + 1] for i in range(op.order)]
    assert [pl.part_mode_k[i] for i in range(op.part_count)] == \
        [(op.part_mode[i] << 5) | op.part_k[i] for i in range(op.part_count)]
    assert pl.payload_bytes == nbytes


@pytest.mark.parametrize("kind", KINDS)
def test_block_plans_match_oracle(gpu, oracle, kind):
    """k_analyze<16,1024> decisions (predictor, coefficients, partition order, modes, k, exact bits)."""
    left, right = gpu.synth.synth_pcm(16384 * 2 + 777, 2, 24, 96000, seed=33, kind=kind)
    m, s = _ms(left, right)
    be = gpu.lacx.BlockEncoder()
    for x in (left[:16384], s[16384:32768], m[32768:], right[3:4100], left[7:40], s[9:40], m[:1], left[:2],
              right[:13], s[1000:1256], left[:13312], m[:12289]):
        pl = be.plan(x)
        _check_plan(pl, oracle.block_plan(x), len(oracle.block_encode(x)))
        assert be.encode(x) == oracle.block_encode(x)


def test_block_flags(gpu, oracle):
    left, right = gpu.synth.synth_pcm(16384, 2, 16, 48000, seed=4, kind="mixed")
    for zr in (False, True):
        for pt in (False, True):
            be = gpu.lacx.BlockEncoder()
            be.set_zero_run_enabled(zr)
            be.set_partitioning_enabled(pt)
            for x in (left, np.zeros(5000, np.int32), (right // 4096).astype(np.int32)):
                assert be.encode(x) == oracle.block_encode(x, zr, pt)
    assert gpu.lacx.BlockEncoder().encode(np.zeros(0, np.int32)) == oracle.block_encode(np.zeros(0, np.int32))


STREAMS = [
    # frames, channels, bit_depth, rate, stereo_mode, kind, stereo
    (16384 * 5 + 37, 2, 16, 48000, 2, "music", "wide"),
    (16384 * 4 + 4000, 2, 24, 96000, 2, "mixed", "wide"),
    (16384 * 3, 2, 16, 44100, 2, "noise", "independent"),
    (16384 * 2 + 1, 2, 24, 192000, 2, "music", "narrow"),
    (16384 * 3 + 100, 2, 16, 48000, 0, "mixed", "wide"),
    (16384 * 3 + 100, 2, 16, 48000, 1, "mixed", "wide"),
    (16384 * 3 + 5000, 1, 16, 48000, 0, "mixed", "wide"),
    (2400, 1, 16, 48000, 0, "tone", "wide"),  # BASELINE config 1 shape
    (1, 2, 16, 48000, 2, "noise", "wide"),
    (31, 2, 24, 48000, 2, "music", "identical"),
    (33, 2, 16, 48000, 2, "music", "anticorr"),
    (257, 2, 16, 48000, 2, "walk", "wide"),
    (4096, 2, 16, 48000, 2, "noise", "independent"),
    (4097, 2, 16, 48000, 2, "noise", "independent"),
    (16384 + 300, 2, 16, 48000, 2, "silence", "identical"),
    (16384 * 2, 2, 24, 48000, 2, "sparse", "half_silent"),
    (16384 * 2 + 9, 2, 16, 96000, 2, "near_silence", "independent"),
]


@pytest.mark.parametrize("host_emit", [False, True], ids=["device_emit", "host_emit"])
@pytest.mark.parametrize("case", STREAMS, ids=lambda c: "-".join(map(str, c)))
def test_stream_bytes_match_oracle(gpu, oracle, case, host_emit):
    frames, ch, bd, sr, sm, kind, st = case
    left, right = gpu.synth.synth_pcm(frames, ch, bd, sr, seed=frames % 97 + 3, kind=kind, stereo=st)
    enc = gpu.lacx.Encoder(12, sm, sr, bd)
    enc.set_host_emit(host_emit)
    got = enc.encode(left, right)
    want = oracle.encode(left, right, sr, bd, sm, threads=8)
    assert got == want
    l2, r2, hdr = oracle.decode(got)
    assert np.array_equal(l2, left)
    if ch == 2:
        assert np.array_equal(r2, right)


def test_stream_flags_and_errors(gpu, oracle):
    left, right = gpu.synth.synth_pcm(16384 * 2 + 50, 2, 16, 48000, seed=8, kind="mixed")
    for zr in (False, True):
        for pt in (False, True):
            enc = gpu.lacx.Encoder(12, 2, 48000, 16)
            enc.set_zero_run_enabled(zr)
            enc.set_partitioning_enabled(pt)
            assert enc.encode(left, right) == oracle.encode(left, right, 48000, 16, 2, zr, pt)
    # error behaviour of LAC::Encoder::encode (ref src/codec/lac/encoder.cpp:220-241)
    with pytest.raises(ValueError, match="left channel must not be empty"):
        gpu.lacx.Encoder(12, 2, 48000, 16).encode(np.zeros(0, np.int32), None)
    with pytest.raises(ValueError, match="must match left channel size"):
        gpu.lacx.Encoder(12, 2, 48000, 16).encode(left, right[:-1])
    with pytest.raises(ValueError, match="unsupported sample rate"):
        gpu.lacx.Encoder(12, 2, 22050, 16).encode(left, right)
    with pytest.raises(ValueError, match="unsupported bit depth"):
        gpu.lacx.Encoder(12, 2, 48000, 20).encode(left, right)
    with pytest.raises(ValueError, match="unsupported stereo mode"):
        gpu.lacx.Encoder(12, 3, 48000, 16).encode(left, right)
    bad = right.copy()
    bad[16384 + 7] = 40000
    with pytest.raises(ValueError, match=r"right sample at index 16391 is outside"):
        gpu.lacx.Encoder(12, 2, 48000, 16).encode(left, bad)
    bad2 = left.copy()
    bad2[20000] = -40000
    with pytest.raises(ValueError, match=r"left sample at index 20000 is outside"):
        gpu.lacx.Encoder(12, 2, 48000, 16).encode(bad2, bad)


def test_shards_concatenate_to_the_whole_stream(gpu, oracle):
    """Block-range split (multi-GPU path): shard payloads + tables assemble to the one-shot bytes."""
    left, right = gpu.synth.synth_pcm(16384 * 7 + 321, 2, 16, 48000, seed=12, kind="mixed")
    whole = gpu.lacx.Encoder(12, 2, 48000, 16).encode(left, right)
    for cuts in ([3 * 16384], [16384, 5 * 16384], [2 * 16384, 4 * 16384, 6 * 16384]):
        bounds = [0] + cuts + [left.size]
        shards = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            shards.append(gpu.lacx.Encoder(12, 2, 48000, 16).encode_shard(left[a:b], right[a:b]))
        assert gpu.lacx.assemble(48000, 16, 2, 2, shards) == whole
    assert whole == oracle.encode(left, right, 48000, 16, 2, threads=8)


def test_golden_digests(gpu):
    """Full-size BASELINE configs against digests minted from the unmodified reference build."""
    path = os.path.join(GOLDEN, "digests.json")
    if not os.path.exists(path):
        pytest.skip("no golden digests committed")
    with open(path) as f:
        entries = json.load(f)
    for ent in entries:
        if ent.get("gpu_test", True) is False or ent["name"].startswith("cfg5_"):
            continue
        g = ent["gen"]
        left, right = gpu.synth.synth_pcm(g["frames"], g["channels"], g["bit_depth"], g["sample_rate"],
                                          seed=g["seed"], kind=g["kind"], stereo=g.get("stereo", "wide"),
                                          start=g.get("start", 0))
        for host_emit in (False, True):
            enc = gpu.lacx.Encoder(12, ent["stereo_mode"], g["sample_rate"], g["bit_depth"])
            enc.set_host_emit(host_emit)
            got = enc.encode(left, right)
            assert len(got) == ent["lac_bytes"], (ent["name"], host_emit)
            assert hashlib.sha256(got).hexdigest() == ent["lac_sha256"], (ent["name"], host_emit)


def test_golden_digests_of_the_mixed_corpus_as_one_batch(gpu):
    """BASELINE configs[4]: the reference-minted cfg5_* digests, their streams encoded as ONE batch job."""
    import torch

    with open(os.path.join(GOLDEN, "digests.json")) as f:
        entries = [e for e in json.load(f) if e["name"].startswith("cfg5_") and e.get("gpu_test", True)]
    assert entries
    streams, keep = [], []
    for ent in entries:
        g = ent["gen"]
        left, right = gpu.synth.synth_pcm(g["frames"], g["channels"], g["bit_depth"], g["sample_rate"], seed=g["seed"],
                                          kind=g["kind"], stereo=g.get("stereo", "wide"))
        inter = gpu.synth.interleave(left, right, g["bit_depth"])
        d = torch.from_numpy(inter.view(np.int16) if g["bit_depth"] == 16 else inter).cuda()
        keep.append(d)
        streams.append((d.data_ptr(), gpu.lacx.PCM_INTERLEAVED_I16 if g["bit_depth"] == 16 else gpu.lacx.PCM_INTERLEAVED_I24,
                        g["channels"], g["frames"]))
    be = gpu.lacx.BatchEncoder([(e["gen"]["sample_rate"], e["gen"]["bit_depth"], e["stereo_mode"]) for e in entries], device=0)
    res = be.encode_device(streams)
    for ent, (pay, tab) in zip(entries, res):
        g = ent["gen"]
        got = gpu.lacx.assemble(g["sample_rate"], g["bit_depth"], ent["stereo_mode"], g["channels"], [(pay.tobytes(), tab.copy())])
        assert len(got) == ent["lac_bytes"] and hashlib.sha256(got).hexdigest() == ent["lac_sha256"], ent["name"]


def test_cpp_mirror_classes_on_device(gpu):
    import subprocess

    from test_host_side import _build_mirror_test

    assert subprocess.call([_build_mirror_test()]) == 0


def test_forced_ms_still_validates_left_right(gpu):
    left, right = gpu.synth.synth_pcm(16384 + 50, 2, 16, 48000, seed=8, kind="music")
    bad = left.copy()
    bad[16400] = 70000
    with pytest.raises(ValueError, match=r"left sample at index 16400 is outside"):
        gpu.lacx.Encoder(12, 1, 48000, 16).encode(bad, right)


@pytest.mark.parametrize("case", [(16384 * 3 + 500, 2, 16, 48000, 2, "mixed"), (16384 * 2 + 77, 1, 16, 44100, 0, "music"),
                                  (16384 * 2 + 4001, 2, 24, 96000, 2, "mixed"), (5000, 1, 24, 48000, 0, "noise"),
                                  (16384 * 2, 2, 16, 48000, 1, "music")])
def test_interleaved_device_ingest(gpu, oracle, case):
    """WAV-layout PCM (interleaved int16 / packed int24) read directly by the kernels."""
    import torch

    frames, ch, bd, sr, sm, kind = case
    left, right = gpu.synth.synth_pcm(frames, ch, bd, sr, seed=17, kind=kind)
    inter = gpu.synth.interleave(left, right, bd)
    d = torch.from_numpy(inter.view(np.int16) if bd == 16 else inter).cuda()
    enc = gpu.lacx.Encoder(12, sm, sr, bd, device=0)
    layout = gpu.lacx.PCM_INTERLEAVED_I16 if bd == 16 else gpu.lacx.PCM_INTERLEAVED_I24
    payload, table = enc.encode_shard_pcm_device_view(d.data_ptr(), layout, ch, frames,
                                                      torch.cuda.current_stream().cuda_stream)
    got = gpu.lacx.assemble(sr, bd, sm, ch, [(payload.tobytes(), table.copy())])
    assert got == oracle.encode(left, right, sr, bd, sm, threads=8)


@pytest.mark.parametrize("bd", [16, 24])
def test_unaligned_device_pointers(gpu, oracle, bd):
    """Device buffers that are not 16-byte aligned take the per-sample staging path: same bytes."""
    import torch

    frames, ch, sr, sm = 16384 * 2 + 123, 2, 48000, 2
    left, right = gpu.synth.synth_pcm(frames, ch, bd, sr, seed=23, kind="mixed")
    want = oracle.encode(left, right, sr, bd, sm, threads=8)
    enc = gpu.lacx.Encoder(12, sm, sr, bd, device=0)
    # interleaved source shifted by one frame inside a larger allocation
    inter = gpu.synth.interleave(left, right, bd)
    flat = inter.view(np.int16).reshape(-1) if bd == 16 else inter.reshape(-1)
    frame_elems = flat.size // frames
    big = torch.zeros(flat.size + frame_elems, dtype=torch.int16 if bd == 16 else torch.uint8, device="cuda")
    big[frame_elems:] = torch.from_numpy(flat.copy()).cuda()
    layout = gpu.lacx.PCM_INTERLEAVED_I16 if bd == 16 else gpu.lacx.PCM_INTERLEAVED_I24
    ptr = big.data_ptr() + frame_elems * big.element_size()
    assert ptr % 16 != 0
    payload, table = enc.encode_shard_pcm_device_view(ptr, layout, ch, frames, torch.cuda.current_stream().cuda_stream)
    assert gpu.lacx.assemble(sr, bd, sm, ch, [(payload.tobytes(), table.copy())]) == want
    # planar int32 shifted by one sample
    dl = torch.zeros(frames + 1, dtype=torch.int32, device="cuda")
    dr = torch.zeros(frames + 1, dtype=torch.int32, device="cuda")
    dl[1:] = torch.from_numpy(left).cuda()
    dr[1:] = torch.from_numpy(right).cuda()
    torch.cuda.synchronize()
    assert enc.encode_device(dl.data_ptr() + 4, dr.data_ptr() + 4, left, right, frames) == want


def test_encode_wav_matches_reference_chain(gpu, oracle):
    """lacx_encode_wav == read_wav + LAC::Encoder::encode (raw data chunk to the device, ingest on load)."""
    import wavutil as W

    l16, r16 = gpu.synth.synth_pcm(16384 * 2 + 999, 2, 16, 48000, seed=31, kind="music")
    l24, _ = gpu.synth.synth_pcm(16384 + 5, 1, 24, 96000, seed=32, kind="mixed")
    odd = W.chunk(b"LIST", b"abc")
    wav = W.make_wav(l16, r16, 48000, 16, before=[odd], after=[odd])
    assert gpu.lacx.Encoder(12, 2, 48000, 16).encode_wav(wav) == oracle.encode(l16, r16, 48000, 16, 2, threads=8)
    wav = W.make_wav(l24, None, 96000, 24, between=[W.chunk(b"fact", b"12345")])
    assert gpu.lacx.Encoder(12, 0, 96000, 24).encode_wav(wav) == oracle.encode(l24, None, 96000, 24, 0, threads=8)
    assert gpu.lacx.Encoder(12, 0, 96000, 24).encode_wav_view(wav).tobytes() == oracle.encode(l24, None, 96000, 24, 0, threads=8)
    # pipelined upload (several chunks) + in-place container
    lbig, rbig = gpu.synth.synth_pcm(16384 * 700 + 33, 2, 16, 48000, seed=33, kind="music")
    wbig = W.make_wav(lbig, rbig, 48000, 16)
    enc = gpu.lacx.Encoder(12, 2, 48000, 16)
    want = oracle.encode(lbig, rbig, 48000, 16, 2, threads=8)
    assert enc.encode_wav_view(wbig).tobytes() == want
    assert enc.encode_wav(wbig) == want
    with pytest.raises(ValueError, match="differs from the encoder"):
        gpu.lacx.Encoder(12, 0, 48000, 24).encode_wav(wav)
    with pytest.raises(ValueError, match="not a supported PCM WAV"):
        gpu.lacx.Encoder(12, 0, 48000, 24).encode_wav(wav[:-1])


def test_config5_every_format_combination(gpu, oracle):
    """BASELINE configs[4]: {mono, stereo} x {16, 24 bit} x {44.1, 48, 96, 192 kHz} as one batched job through one
    process: per-block predictor choice + stereo auto-select for every combination, bytes equal to the oracle's."""
    seed = 100
    for ch in (1, 2):
        for bd in (16, 24):
            for sr in (44100, 48000, 96000, 192000):
                seed += 1
                frames = sr * 2 + 321
                left, right = gpu.synth.synth_pcm(frames, ch, bd, sr, seed=seed, kind="mixed" if seed & 1 else "music")
                sm = 2 if ch == 2 else 0
                got = gpu.lacx.Encoder(12, sm, sr, bd).encode(left, right)
                assert got == oracle.encode(left, right, sr, bd, sm, threads=8), (ch, bd, sr)


def test_random_sweep(gpu, oracle):
    """Randomised shapes, material, levels and flags (a short run of scripts/fuzz_blocks.py's sweep)."""
    rng = np.random.default_rng(20261003)
    kinds = ["music", "noise", "silence", "near_silence", "sparse", "ramp", "walk", "tone", "mixed"]
    stereos = ["wide", "narrow", "identical", "anticorr", "independent"]
    for it in range(60):
        ch = int(rng.integers(1, 3))
        bd = int(rng.choice([16, 24]))
        sr = int(rng.choice([44100, 48000, 96000, 192000]))
        sm = int(rng.integers(0, 3)) if ch == 2 else 0
        frames = int(rng.choice([1, 33, 4095, 4097, 16383, 16385, int(rng.integers(1, 50000))]))
        left, right = gpu.synth.synth_pcm(frames, ch, bd, sr, seed=int(rng.integers(1, 10**6)), kind=str(rng.choice(kinds)),
                                          stereo=str(rng.choice(stereos)), start=int(rng.integers(0, 10**6)))
        if rng.random() < 0.3:
            sh = int(rng.integers(1, bd - 1))
            left = (left >> sh).astype(np.int32)
            right = None if right is None else (right >> sh).astype(np.int32)
        zr, pt = bool(rng.random() < 0.85), bool(rng.random() < 0.85)
        enc = gpu.lacx.Encoder(12, sm, sr, bd)
        enc.set_zero_run_enabled(zr)
        enc.set_partitioning_enabled(pt)
        enc.set_host_emit(bool(rng.random() < 0.3))
        assert enc.encode(left, right) == oracle.encode(left, right, sr, bd, sm, zero_run=zr, partitioning=pt, threads=8), \
            (it, ch, bd, sr, sm, frames, zr, pt)
    be = gpu.lacx.BlockEncoder()
    for it in range(200):
        n = int(rng.choice([1, 2, 13, 31, 32, 33, 255, 256, 257, 4096, 16384, int(rng.integers(1, 16385))]))
        x, _ = gpu.synth.synth_pcm(n, 1, 24, 48000, seed=int(rng.integers(1, 10**6)), kind=str(rng.choice(kinds)),
                                   start=int(rng.integers(0, 10**6)))
        if rng.random() < 0.4:
            x = (x >> int(rng.integers(1, 23))).astype(np.int32)
        assert be.encode(x) == oracle.block_encode(x), (it, n)


def _punch_gaps(rng, x):
    x = x.copy()
    pos = 0
    while pos < x.size:
        pos += int(rng.choice([1, 1, 2, 3, 7, 16, 40, 300]))
        gap = int(rng.choice([1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 1000, 5000]))
        x[pos:pos + gap] = 0
        pos += gap
    return x


def test_zero_run_structures(gpu, oracle):
    """Zero runs of every length class, inside and across the threads' 16-sample chunks, all-zero and almost-all-zero
    blocks: the cases where the candidate pruning bound leans on its zero-run term (count of zeros, run ends)."""
    rng = np.random.default_rng(77)
    be = gpu.lacx.BlockEncoder()
    blocks = [np.zeros(16384, np.int32), np.zeros(4097, np.int32), np.zeros(5, np.int32)]
    one = np.zeros(16384, np.int32)
    one[8000] = 1
    blocks.append(one)
    ends = np.zeros(16384, np.int32)
    ends[0] = ends[-1] = -3
    blocks.append(ends)
    for period in (2, 4, 5, 16, 17, 64):  # one non-zero sample every `period`
        x = np.zeros(16384, np.int32)
        x[::period] = rng.integers(-4, 5, size=x[::period].size)
        blocks.append(x)
    for it in range(40):
        n = int(rng.choice([257, 4096, 16384, int(rng.integers(300, 16385))]))
        x, _ = gpu.synth.synth_pcm(n, 1, 24, 48000, seed=int(rng.integers(1, 10**6)), kind=str(rng.choice(["music", "noise", "walk", "sparse"])))
        if rng.random() < 0.5:
            x = (x >> int(rng.integers(8, 23))).astype(np.int32)
        blocks.append(_punch_gaps(rng, x))
    for i, x in enumerate(blocks):
        assert be.encode(x) == oracle.block_encode(x), i
    for it in range(6):
        frames = int(rng.integers(20000, 60000))
        bd = int(rng.choice([16, 24]))
        left, right = gpu.synth.synth_pcm(frames, 2, bd, 48000, seed=int(rng.integers(1, 10**6)), kind="music", stereo="wide")
        left, right = _punch_gaps(rng, left), _punch_gaps(rng, right)
        if it & 1:
            right = left.copy()  # side channel all zero
        assert gpu.lacx.Encoder(12, 2, 48000, bd).encode(left, right) == oracle.encode(left, right, 48000, bd, 2, threads=8), it


def test_loud_blocks_around_the_32_bit_sum_limit(gpu, oracle):
    """Residual sums in [2^31, 2^32) and on either side of kNarrowLimit (analyze_core.h): the 32-bit paths of the analysis,
    the partition search and the emit up to the limit, the 64-bit ones beyond -- blocks through Block::Encoder, then the
    same material as a 24-bit stereo stream (probe, stereo decision, fused emit)."""
    rng = np.random.default_rng(41)
    be = gpu.lacx.BlockEncoder()
    tone, _ = gpu.synth.synth_pcm(16384, 1, 24, 96000, seed=12, kind="music")
    blocks = [rng.integers(-amp, amp + 1, size=16384).astype(np.int32)
              for amp in (140_000, 185_000, 250_000, 261_000, 262_100, 263_500, 275_000, 600_000)]
    for gain in (6, 14, 30):
        loud = np.clip(tone.astype(np.int64) * gain, -(1 << 23), (1 << 23) - 1).astype(np.int32)
        blocks.append((loud + rng.integers(-150_000, 150_001, size=loud.size)).astype(np.int32))
    blocks.append(rng.integers(-200_000, 200_001, size=9000).astype(np.int32))
    for i, x in enumerate(blocks):
        assert be.encode(x) == oracle.block_encode(x), i
    lim = (1 << 23) - 1
    left = np.clip(np.concatenate(blocks[:6]), -lim - 1, lim).astype(np.int32)
    right = np.clip(np.concatenate(blocks[5:11]), -lim - 1, lim).astype(np.int32)[:left.size]
    left = left[:right.size]
    for sm in (0, 1, 2):
        assert gpu.lacx.Encoder(12, sm, 96000, 24).encode(left, right) == oracle.encode(left, right, 96000, 24, sm, threads=8), sm


def test_begin_end_interface(gpu, oracle):
    """The two-halves interface (enqueue, collect later) with two encoders alternating: same bytes."""
    import torch

    sr, bd, sm = 48000, 16, 2
    streams = []
    for seed in (41, 42, 43):
        left, right = gpu.synth.synth_pcm(16384 * 3 + 1000 * seed % 7000 + 5, 2, bd, sr, seed=seed, kind="mixed")
        d = torch.from_numpy(gpu.synth.interleave(left, right, bd).view(np.int16)).cuda()
        streams.append((left, right, d))
    encs = [gpu.lacx.Encoder(12, sm, sr, bd), gpu.lacx.Encoder(12, sm, sr, bd)]
    got = []
    for i, (left, right, d) in enumerate(streams):
        encs[i % 2].encode_shard_pcm_device_begin(d.data_ptr(), gpu.lacx.PCM_INTERLEAVED_I16, 2, left.size)
        if i >= 1:
            p, t = encs[(i - 1) % 2].encode_shard_end()
            got.append((p.tobytes(), t.copy()))
    p, t = encs[(len(streams) - 1) % 2].encode_shard_end()
    got.append((p.tobytes(), t.copy()))
    for (left, right, _), (p, t) in zip(streams, got):
        assert gpu.lacx.assemble(sr, bd, sm, 2, [(p, t)]) == oracle.encode(left, right, sr, bd, sm, threads=8)
    with pytest.raises(RuntimeError, match="no encode in flight"):
        encs[0].encode_shard_end()


def test_cli_encode_command(gpu, oracle, tmp_path):
    """lacx_cli encode: the reference CLI's encode command (ref src/main.cpp:609-710) over the C ABI."""
    import subprocess

    import wavutil as W

    pkg_dir = os.path.join(ROOT, "lossless-audio-codec_amd")
    subprocess.check_call(["make", "-C", pkg_dir, "lacx_cli"], stdout=subprocess.DEVNULL)
    cli = os.path.join(pkg_dir, "lacx_cli")
    left, right = gpu.synth.synth_pcm(16384 * 2 + 4321, 2, 24, 96000, seed=51, kind="mixed")
    wav = tmp_path / "in.wav"
    wav.write_bytes(W.make_wav(left, right, 96000, 24, before=[W.chunk(b"LIST", b"abc")]))
    for flags, sm, part in (([], 2, True), (["--stereo-mode=ms", "--threads=3"], 1, True), (["--no-partitioning", "--stereo-mode=lr"], 0, False)):
        out = tmp_path / "out.lac"
        res = subprocess.run([cli, "encode", str(wav), str(out)] + flags, capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        want = oracle.encode(left, right, 96000, 24, sm, partitioning=part, threads=8)
        assert out.read_bytes() == want
        assert f"({len(want)} bytes)" in res.stdout
    bad = subprocess.run([cli, "encode", str(wav), str(tmp_path / "x.lac"), "--threads=0"], capture_output=True, text=True)
    assert bad.returncode == 1 and "--threads requires a positive integer" in bad.stderr
    bad = subprocess.run([cli, "encode", str(tmp_path / "missing.wav"), str(tmp_path / "x.lac")], capture_output=True, text=True)
    assert bad.returncode == 1 and "Failed to read WAV" in bad.stderr
    assert subprocess.run([cli, "encode", str(wav), str(wav)], capture_output=True).returncode == 1
    # same file under another spelling / through a symlink (ref src/main.cpp:433-444), input must survive
    link = tmp_path / "alias.wav"
    os.symlink(wav, link)
    before = wav.read_bytes()
    for other in (str(tmp_path / "." / "in.wav"), str(link)):
        res = subprocess.run([cli, "encode", str(wav), other], capture_output=True, text=True)
        assert res.returncode == 1 and "Input and output paths must be different" in res.stderr
    assert wav.read_bytes() == before
    # the reference's --debug-* flags are accepted; LAC_THREADS is resolved by the tool (ref :586-591, thread_limit.hpp)
    env = dict(os.environ, LAC_THREADS="3")
    res = subprocess.run([cli, "encode", str(wav), str(tmp_path / "d.lac"), "--debug-threads", "--debug-lpc", "--debug-stereo-est",
                          "--debug-zr", "--debug-partitions"], capture_output=True, text=True, env=env)
    assert res.returncode == 0, res.stderr
    assert (tmp_path / "d.lac").read_bytes() == oracle.encode(left, right, 96000, 24, 2, threads=8)
    base = oracle.encode(left, right, 96000, 24, 2, zero_run=False, threads=8)
    assert f"[debug-zr] baseline_bytes={len(base)} zr_bytes={len((tmp_path / 'd.lac').read_bytes())}" in res.stdout
    assert "Thread usage: 1 threads" in res.stdout
    for bad_env in ("0", "x3", "-1"):
        res = subprocess.run([cli, "encode", str(wav), str(tmp_path / "e.lac")], capture_output=True, text=True,
                             env=dict(os.environ, LAC_THREADS=bad_env))
        assert res.returncode == 1 and "LAC_THREADS must be a positive integer" in res.stderr
    assert subprocess.run([cli, "encode", str(wav), str(tmp_path / "f.lac"), "--bogus"], capture_output=True).returncode == 1


def test_two_encoders_on_two_host_threads(gpu, oracle):
    """Two encoders whose first launches come from two host threads at once (the per-device kernel attribute state is
    shared and mutex-protected): same bytes as the oracle from both."""
    import threading

    cases = []
    for seed in (61, 62):
        left, right = gpu.synth.synth_pcm(16384 * 4 + 100 * seed, 2, 16, 48000, seed=seed, kind="mixed")
        cases.append((left, right, oracle.encode(left, right, 48000, 16, 2, threads=8)))
    got = [None, None]

    def run(i):
        try:
            enc = gpu.lacx.Encoder(12, 2, 48000, 16, device=0)
            for _ in range(3):
                got[i] = enc.encode(cases[i][0], cases[i][1])
        except Exception as ex:  # noqa: BLE001
            got[i] = ex

    ths = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for i in range(2):
        assert got[i] == cases[i][2], got[i] if isinstance(got[i], Exception) else "bytes differ"


def test_two_devices_in_one_process(gpu, oracle):
    """One encoder per GPU in one process (lacx_config.device): the > 64 KiB dynamic-LDS opt-in is per device."""
    if gpu.lacx.device_count() < 2:
        pytest.skip("needs two HIP devices")
    left, right = gpu.synth.synth_pcm(16384 * 3 + 77, 2, 16, 48000, seed=63, kind="mixed")
    want = oracle.encode(left, right, 48000, 16, 2, threads=8)
    for dev in (1, 0, 1):
        assert gpu.lacx.Encoder(12, 2, 48000, 16, device=dev).encode(left, right) == want


def test_pinned_reservation_regrows(gpu, oracle, monkeypatch):
    """A result larger than the pinned reservation (forced tiny here) is re-emitted into a regrown buffer, not an
    error: every device-emit entry point still returns the reference's bytes."""
    import wavutil as W

    left, right = gpu.synth.synth_pcm(16384 * 6 + 1234, 2, 16, 48000, seed=64, kind="noise", stereo="independent")
    want = oracle.encode(left, right, 48000, 16, 2, threads=8)
    monkeypatch.setenv("LACX_PINNED_CAP_BYTES", "100000")  # the stream needs about 0.4 MB
    enc = gpu.lacx.Encoder(12, 2, 48000, 16, device=0)
    assert enc.encode(left, right) == want
    assert enc.timing().regrows == 1
    assert enc.encode_wav(W.make_wav(left, right, 48000, 16)) == want
    assert enc.timing().regrows == 1
    payload, table = enc.encode_shard(left, right)
    assert gpu.lacx.assemble(48000, 16, 2, 2, [(payload, table)]) == want
    monkeypatch.delenv("LACX_PINNED_CAP_BYTES")
    enc = gpu.lacx.Encoder(12, 2, 48000, 16, device=0)  # (knobs are read when the encoder is created)
    assert enc.encode(left, right) == want
    assert enc.timing().regrows == 0


def test_host_emit_waits_for_the_callers_stream(gpu, oracle):
    """Host-emit pipeline with PCM produced asynchronously on the caller's stream: chunks that run on the encoder's own
    streams must be ordered behind it (>= 384 blocks, so that the pipeline has more than one chunk)."""
    import torch

    frames = 16384 * 400 + 5
    left, right = gpu.synth.synth_pcm(frames, 2, 16, 48000, seed=65, kind="music")
    want = oracle.encode(left, right, 48000, 16, 2, threads=8)
    hl, hr = torch.from_numpy(left).pin_memory(), torch.from_numpy(right).pin_memory()
    side = torch.cuda.Stream()
    enc = gpu.lacx.Encoder(12, 2, 48000, 16, device=0)
    enc.set_host_emit(True)
    for _ in range(2):
        dl = torch.zeros(frames, dtype=torch.int32, device="cuda")
        dr = torch.zeros(frames, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            # a long fill first, so that the copies are still pending when the encode is enqueued
            junk = torch.empty(1 << 28, dtype=torch.int32, device="cuda")
            for _k in range(4):
                junk.fill_(_k)
            dl.copy_(hl, non_blocking=True)
            dr.copy_(hr, non_blocking=True)
            got = enc.encode_device(dl.data_ptr(), dr.data_ptr(), left, right, frames, side.cuda_stream)
        assert got == want
        del junk


@pytest.mark.parametrize("mode", ["fused", "k_emit_only", "every_fifth_left_to_k_emit", "packer_gives_up",
                                  "packer_16_waves", "packer_48_waves"])
def test_fused_emit_and_its_fallbacks(gpu, oracle, monkeypatch, request, mode):
    """The emit fused into the analysis kernel + the streaming packer beside it (default), k_offsets + k_emit alone,
    and the two repair paths: a test hook leaves every fifth channel block to k_emit, another one fills the staging
    slots but never announces them, so that the packer gives up and k_pack moves everything.  Same bytes every time."""
    if mode == "k_emit_only":
        monkeypatch.setenv("LACX_FUSED_EMIT", "0")
    if mode in ("every_fifth_left_to_k_emit", "packer_gives_up"):
        # the hooks exist only in the diagnostic twin of the library (analysis kernel built with -DLACX_TEST_HOOKS)
        gpu.lacx.use_library(gpu.lacx.HOOKS_LIB_PATH)
        request.addfinalizer(lambda: gpu.lacx.use_library(None))
        monkeypatch.setenv("LACX_DEBUG_SKIP", "1024" if mode == "every_fifth_left_to_k_emit" else "8192")
    if mode == "packer_16_waves":  # the packer's waves are independent: any number of them gives the same bytes
        monkeypatch.setenv("LACX_PACK_GRID", "1")
    if mode == "packer_48_waves":
        monkeypatch.setenv("LACX_PACK_GRID", "3")
    cases = [(16384 * 40 + 321, 2, 16, 48000, 2, "mixed"), (16384 * 9 + 4000, 2, 24, 96000, 2, "mixed"),
             (16384 * 7 + 5, 1, 16, 44100, 0, "music"), (16384 * 6, 2, 16, 48000, 1, "music"),
             (16384 * 5 + 77, 2, 24, 48000, 0, "noise"), (16384 * 400 + 9, 2, 16, 48000, 2, "music")]
    for frames, ch, bd, sr, sm, kind in cases:
        left, right = gpu.synth.synth_pcm(frames, ch, bd, sr, seed=71, kind=kind)
        enc = gpu.lacx.Encoder(12, sm, sr, bd, device=0)
        want = oracle.encode(left, right, sr, bd, sm, threads=8)
        for _ in range(2):
            assert enc.encode(left, right) == want, (mode, frames, ch, bd)
        if mode in ("fused", "packer_16_waves", "packer_48_waves"):
            t = enc.timing()
            small_last = ch == 2 and sm == 2 and (frames % 16384) and (frames % 16384) <= 4096
            assert t.emit_direct == (-(-frames // 16384) - (1 if small_last else 0)) * ch
        if mode == "every_fifth_left_to_k_emit":
            assert 0 < enc.timing().emit_direct < -(-frames // 16384) * ch or frames < 16384 * 3
        if mode == "packer_gives_up":
            assert enc.timing().packer_gave_up > 0 and enc.timing().moved_by_k_pack > 0
        enc.close()


def test_batch_of_streams_is_one_job_with_the_same_bytes(gpu, oracle):
    """lacx_encode_batch_device: streams of different rate / depth / channels / stereo mode as ONE launch set (BASELINE
    configs[4]).  Every stream's (payload, table) must assemble to the bytes the oracle gives for that stream alone --
    including streams of one block, ragged final blocks, a final block of <= 4096 frames that is encoded both ways and
    compared (left to the repair emit), forced LR / MS, and mono."""
    import torch

    specs = [
        # frames, channels, bit_depth, rate, stereo_mode, kind
        (16384 * 3 + 500, 2, 16, 48000, 2, "mixed"),
        (16384 * 2 + 77, 1, 16, 44100, 0, "music"),
        (16384 * 2 + 4001, 2, 24, 96000, 2, "mixed"),     # final block <= 4096 frames: both ways
        (5000, 1, 24, 48000, 0, "noise"),
        (16384 * 2, 2, 16, 48000, 1, "music"),
        (300, 2, 16, 192000, 2, "noise"),                  # one small block, both ways
        (16384 * 4 + 12000, 2, 24, 192000, 0, "music"),
        (16384, 2, 16, 96000, 2, "silence"),
        (16384 * 2 + 9, 2, 16, 44100, 2, "near_silence"),
    ]
    streams, keep, want = [], [], []
    for i, (frames, ch, bd, sr, sm, kind) in enumerate(specs):
        left, right = gpu.synth.synth_pcm(frames, ch, bd, sr, seed=40 + i, kind=kind)
        inter = gpu.synth.interleave(left, right, bd)
        d = torch.from_numpy(inter.view(np.int16) if bd == 16 else inter).cuda()
        keep.append(d)
        streams.append((d.data_ptr(), gpu.lacx.PCM_INTERLEAVED_I16 if bd == 16 else gpu.lacx.PCM_INTERLEAVED_I24, ch, frames))
        want.append(oracle.encode(left, right, sr, bd, sm if ch == 2 else 0, threads=8))
    be = gpu.lacx.BatchEncoder([(sr, bd, sm if ch == 2 else 0) for (_, ch, bd, sr, sm, _) in specs], device=0)
    for rep in range(2):  # the second call reuses every buffer
        res = be.encode_device(streams, torch.cuda.current_stream().cuda_stream)
        for (frames, ch, bd, sr, sm, _), (pay, tab), w in zip(specs, res, want):
            got = gpu.lacx.assemble(sr, bd, sm if ch == 2 else 0, ch, [(pay.tobytes(), tab.copy())])
            assert got == w, (frames, ch, bd, sr, sm, rep)
    t = be.timing()
    assert t.packer_gave_up == 0
    # a sample outside the bit depth names its stream (planar int32 input can hold one)
    bad_l, bad_r = gpu.synth.synth_pcm(20000, 2, 16, 48000, seed=3, kind="music")
    bad_l = bad_l.copy()
    bad_l[17000] = 40000
    dl, dr = torch.from_numpy(bad_l).cuda(), torch.from_numpy(bad_r).cuda()
    be2 = gpu.lacx.BatchEncoder([(sr, bd, sm if ch == 2 else 0) for (_, ch, bd, sr, sm, _) in specs[:1]] + [(48000, 16, 2)], device=0)
    with pytest.raises(ValueError, match=r"stream 1: left sample at index 17000 is outside"):
        be2.encode_device(streams[:1] + [(dl.data_ptr(), gpu.lacx.PCM_PLANAR_I32, 2, 20000, dr.data_ptr())])


def test_block_encoder_full_int32_domain(gpu, oracle):
    """Block::Encoder::encode takes any int32 samples (ref block/encoder.cpp:313-316): outside the 25-bit domain of
    validated input the LPC residual can leave int32 and the reference falls back through the orders {12, 10, 8, 6, 4}
    to none (ref lpc.cpp:24-36, 188-229), zigzag values use all 32 bits and k saturates at 31.  The wide kernel
    (csrc/wide.hip) must give the reference's bytes there (oracle: lac_oracle.c:291-343 restates the fallback)."""
    rng = np.random.default_rng(77)
    be = gpu.lacx.BlockEncoder(12)
    cases = []
    n = 4096
    t = np.arange(n)
    # 26..31-bit material: smooth (LPC wins, residuals small), noisy, and mixtures with full-scale steps
    for bits in (26, 28, 30, 31):
        amp = (1 << (bits - 1)) - 1
        cases.append((np.sin(t / 37.0) * amp * 0.9).astype(np.int64))
        cases.append(rng.integers(-amp, amp, size=n))
        sq = np.where((t // 64) % 2 == 0, amp, -amp).astype(np.int64)
        cases.append(sq + rng.integers(-1000, 1000, size=n))
    # alternating extremes: every difference overflows int32, every LPC order has to fall back
    alt = np.where(t % 2 == 0, 2**31 - 1, -2**31).astype(np.int64)
    cases.append(alt)
    cases.append(np.concatenate([alt[:2000], np.zeros(2096, np.int64)]))
    # a block that is inside the domain except for one sample
    one = (np.sin(t / 11.0) * 30000).astype(np.int64)
    one[1234] = 2**29
    cases.append(one)
    # short blocks and a full-size one
    cases.append(rng.integers(-2**30, 2**30, size=37))
    cases.append(rng.integers(-2**27, 2**27, size=300))
    cases.append((np.sin(np.arange(16384) / 90.0) * (2**29)).astype(np.int64) + rng.integers(-2**20, 2**20, size=16384))
    for i, x in enumerate(cases):
        x = np.clip(x, -2**31, 2**31 - 1).astype(np.int32)
        for zr, pt in ((True, True), (False, False)):
            be.set_zero_run_enabled(zr)
            be.set_partitioning_enabled(pt)
            got = be.encode(x)
            want = oracle.block_encode(x, zr, pt)
            assert got == want, (i, zr, pt, len(got), len(want))
    with pytest.raises(ValueError, match="larger than 16384"):
        be.encode(np.zeros(16385, np.int32))


def test_one_handle_through_batch_shard_batch_wide_block(gpu, oracle):
    """One encoder handle used for a batch, then a drained shard encode, then a larger batch (the descriptor table grows),
    then a wide Block::Encoder block: the batch's growth path must leave the shard path's buffers and streams alone
    (round-3 advisor finding: it freed them)."""
    import torch

    def stream(frames, seed, kind="mixed"):
        left, right = gpu.synth.synth_pcm(frames, 2, 16, 48000, seed=seed, kind=kind)
        inter = gpu.synth.interleave(left, right, 16)
        return left, right, torch.from_numpy(inter.view(np.int16)).cuda()

    a = [stream(16384 * 3 + 10 * i, 100 + i) for i in range(2)]
    b = [stream(16384 * 2 + 7 * i, 200 + i) for i in range(40)]  # 40 streams: the descriptor table must grow
    be = gpu.lacx.BatchEncoder([(48000, 16, 2)] * len(a), device=0)
    enc = be._enc  # the handle under test

    def check_batch(be_, set_):
        res = be_.encode_device([(d.data_ptr(), gpu.lacx.PCM_INTERLEAVED_I16, 2, l.size) for l, r, d in set_])
        for (l, r, d), (pay, tab) in zip(set_, res):
            assert gpu.lacx.assemble(48000, 16, 2, 2, [(pay.tobytes(), tab.copy())]) == oracle.encode(l, r, 48000, 16, 2, threads=8)

    check_batch(be, a)
    l, r, d = stream(16384 * 300 + 55, 300, "music")  # more than one packer range
    want = oracle.encode(l, r, 48000, 16, 2, threads=8)
    for _ in range(2):
        pay, tab = enc.encode_shard_pcm_device_view(d.data_ptr(), gpu.lacx.PCM_INTERLEAVED_I16, 2, l.size)
        assert gpu.lacx.assemble(48000, 16, 2, 2, [(pay.tobytes(), np.array(tab, dtype=np.uint32))]) == want
    be.formats = [(48000, 16, 2)] * len(b)
    check_batch(be, b)
    pay, tab = enc.encode_shard_pcm_device_view(d.data_ptr(), gpu.lacx.PCM_INTERLEAVED_I16, 2, l.size)
    assert gpu.lacx.assemble(48000, 16, 2, 2, [(pay.tobytes(), np.array(tab, dtype=np.uint32))]) == want
    # a wide block on the same handle (lacx_block_encode's own scratch), before and after another batch
    wide = (np.arange(5000, dtype=np.int64) * 1000003 % (1 << 31) - (1 << 30)).astype(np.int32)
    import ctypes as C
    for _ in range(2):
        out, size = C.POINTER(C.c_uint8)(), C.c_uint64()
        rc = gpu.lacx.lib().lacx_block_encode(enc._handle(), wide.ctypes.data_as(C.POINTER(C.c_int32)), C.c_uint32(wide.size),
                                              C.byref(out), C.byref(size))
        assert rc == 0
        got = C.string_at(out, size.value)
        gpu.lacx.lib().lacx_free(out)
        assert got == oracle.block_encode(wide)
        check_batch(be, b)


def test_batch_regrows_instead_of_failing(gpu, oracle, monkeypatch):
    """A stream of a batch that needs more than its estimated pinned region (forced tiny here) makes the job run once more
    with exact regions -- the reference never fails on size, and neither does the shard path."""
    import torch

    monkeypatch.setenv("LACX_PINNED_CAP_BYTES", "20000")
    specs = [(16384 * 3 + 500, 2, 16, 48000, 2, "noise"), (16384 * 2 + 77, 1, 24, 44100, 0, "noise"), (900, 2, 16, 96000, 2, "music")]
    streams, keep, want = [], [], []
    for i, (frames, ch, bd, sr, sm, kind) in enumerate(specs):
        left, right = gpu.synth.synth_pcm(frames, ch, bd, sr, seed=60 + i, kind=kind)
        inter = gpu.synth.interleave(left, right, bd)
        d = torch.from_numpy(inter.view(np.int16) if bd == 16 else inter).cuda()
        keep.append(d)
        streams.append((d.data_ptr(), gpu.lacx.PCM_INTERLEAVED_I16 if bd == 16 else gpu.lacx.PCM_INTERLEAVED_I24, ch, frames))
        want.append(oracle.encode(left, right, sr, bd, sm, threads=8))
    be = gpu.lacx.BatchEncoder([(sr, bd, sm) for (_, _, bd, sr, sm, _) in specs], device=0)
    for _ in range(2):
        res = be.encode_device(streams)
        for (frames, ch, bd, sr, sm, _), (pay, tab), w in zip(specs, res, want):
            assert gpu.lacx.assemble(sr, bd, sm, ch, [(pay.tobytes(), tab.copy())]) == w
        assert be.timing().regrows == 1


@pytest.mark.parametrize("template", [True, False], ids=["copies", "every_slot_analysed"])
def test_silent_blocks_are_copies_of_the_first(gpu, oracle, monkeypatch, template):
    """Channel blocks of nothing but zeros: the first workgroup that finishes one leaves plan and bitstream behind, later
    ones copy them (lacx_timing.silent_copies counts them; LACX_NO_SILENT_TEMPLATE analyses every slot).  Same bytes as the
    oracle either way -- silence between music, a silent side channel, a silent ragged last block (its length differs:
    never a copy), zero-run coding and partitioning switched off, mono, batch of two formats."""
    if not template:
        monkeypatch.setenv("LACX_NO_SILENT_TEMPLATE", "1")
    rng = np.random.default_rng(2031)
    B = 16384
    # (blocks, channels, bit depth, stereo mode, zero_run, partitioning, frames of the ragged tail)
    cases = [(700, 2, 16, 2, True, True, 777), (420, 2, 24, 0, True, True, 0), (300, 1, 16, 0, True, True, 4000),
             (330, 2, 16, 1, False, True, 5), (330, 2, 24, 2, True, False, 16383), (300, 2, 16, 2, False, False, 0)]
    for nblk, ch, bd, sm, zr, part, tail in cases:
        frames = nblk * B + tail
        left = np.zeros(frames, np.int32)
        right = np.zeros(frames, np.int32) if ch == 2 else None
        music_l, music_r = gpu.synth.synth_pcm(B * 12, 2, bd, 48000, seed=int(rng.integers(1, 10**6)), kind="music", stereo="wide")
        for j, b in enumerate(sorted(rng.choice(nblk, size=12, replace=False))):  # twelve blocks of music in the silence
            left[b * B:(b + 1) * B] = music_l[j * B:(j + 1) * B]
            if ch == 2 and j % 3:  # (every third one with a silent right channel / an all-zero difference)
                right[b * B:(b + 1) * B] = music_r[j * B:(j + 1) * B] if j % 3 == 1 else left[b * B:(b + 1) * B]
        left[int(rng.integers(0, frames))] = 1  # and one block that is silent but for one sample
        enc = gpu.lacx.Encoder(12, sm, 48000, bd, device=0)
        enc.set_zero_run_enabled(zr)
        enc.set_partitioning_enabled(part)
        want = oracle.encode(left, right, 48000, bd, sm, zero_run=zr, partitioning=part, threads=8)
        for _ in range(2):
            assert enc.encode(left, right) == want, (nblk, ch, bd, sm, zr, part, tail)
            copies = enc.timing().silent_copies
            if template:
                assert copies >= (nblk - 14) * ch - 300, copies  # all but the ones the 256 workgroups start with
            else:
                assert copies == 0
        enc.close()
    # two streams of different formats as one job: the silent slots of both are copies of the same block
    import torch

    specs = [(200 * B + 9, 2, 16, 48000, 2), (150 * B, 1, 24, 96000, 0)]
    streams, keep, want = [], [], []
    for frames, ch, bd, sr, sm in specs:
        left = np.zeros(frames, np.int32)
        right = np.zeros(frames, np.int32) if ch == 2 else None
        left[5 * B + 3] = -7
        inter = gpu.synth.interleave(left, right, bd)
        d = torch.from_numpy(inter.view(np.int16) if bd == 16 else inter).cuda()
        keep.append(d)
        streams.append((d.data_ptr(), gpu.lacx.PCM_INTERLEAVED_I16 if bd == 16 else gpu.lacx.PCM_INTERLEAVED_I24, ch, frames))
        want.append(oracle.encode(left, right, sr, bd, sm, threads=8))
    be = gpu.lacx.BatchEncoder([(sr, bd, sm) for (_, _, bd, sr, sm) in specs], device=0)
    res = be.encode_device(streams, torch.cuda.current_stream().cuda_stream)
    for (frames, ch, bd, sr, sm), (pay, tab), w in zip(specs, res, want):
        assert gpu.lacx.assemble(sr, bd, sm, ch, [(pay.tobytes(), tab.copy())]) == w, (frames, ch, bd)
    assert (be.timing().silent_copies > 0) == template


@pytest.mark.parametrize("fold", [True, False], ids=["last_workgroup_decides", "k_stereo_and_k_decide"])
def test_stereo_estimate_and_choice_inside_or_beside_their_producers(gpu, oracle, monkeypatch, fold):
    """The block's stereo estimate is made by the last of its four ingest workgroups and its LR/MS choice by the last of its
    twelve probe slots (default), or by the kernels k_stereo / k_decide (LACX_NO_FRONT_FOLD): same bytes, twice in a row on
    one handle (the counters reset themselves), for auto / forced stereo, mono, a ragged and a small final block, and a
    shard long enough for the two-halves launch of the front kernels."""
    if not fold:
        monkeypatch.setenv("LACX_NO_FRONT_FOLD", "1")
    cases = [(16384 * 30 + 321, 2, 16, 48000, 2, "mixed"), (16384 * 9 + 4000, 2, 24, 96000, 2, "mixed"),
             (16384 * 7 + 5, 1, 16, 44100, 0, "music"), (16384 * 6, 2, 16, 48000, 1, "music"),
             (16384 * 5 + 77, 2, 24, 48000, 0, "noise"), (16384 * 1100 + 9, 2, 16, 48000, 2, "silence")]
    for frames, ch, bd, sr, sm, kind in cases:
        left, right = gpu.synth.synth_pcm(frames, ch, bd, sr, seed=91, kind=kind)
        if kind == "silence":  # (cheap for the oracle; some blocks that are probed in between)
            l2, r2 = gpu.synth.synth_pcm(16384 * 8, 2, bd, sr, seed=92, kind="noise")
            left[16384 * 500:16384 * 508] = l2
            right[16384 * 500:16384 * 508] = r2
        enc = gpu.lacx.Encoder(12, sm, sr, bd, device=0)
        want = oracle.encode(left, right, sr, bd, sm, threads=8)
        for _ in range(2):
            assert enc.encode(left, right) == want, (fold, frames, ch, bd)
        import torch
        if ch == 2:  # and resident in HBM: one chunk, persistent analysis, front kernels in two halves
            inter = gpu.synth.interleave(left, right, bd)
            d = torch.from_numpy(inter.view(np.int16) if bd == 16 else inter).cuda()
            layout = gpu.lacx.PCM_INTERLEAVED_I16 if bd == 16 else gpu.lacx.PCM_INTERLEAVED_I24
            for _ in range(2):
                pay, tab = enc.encode_shard_pcm_device_view(d.data_ptr(), layout, 2, frames, 0)
                got = gpu.lacx.assemble(sr, bd, sm, 2, [(pay.tobytes(), np.array(tab, dtype=np.uint32))])
                assert got == want, (fold, frames, "device view")
        enc.close()
