"""The device decoder (include/lacx.h lacx_decode) on the MI355X: reference-minted .lac files must
give back the PCM they were made from, every stream the device encoder produces must decode to its input and to what the
oracle's decoder (pinned against the reference's) makes of it, and damaged streams must be refused, not mis-decoded."""
import hashlib
import json
import os

import numpy as np
import pytest

import __graft_entry__ as ge
import oracleshim

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gpu():
    pkg = ge.load_pkg()
    if pkg.lacx.device_count() <= 0:
        pytest.fail("no HIP device: the decoder has no CPU fallback")
    return pkg


@pytest.fixture(scope="module")
def oracle():
    return oracleshim


def test_reference_minted_files_decode_to_their_pcm(gpu):
    """tests/golden/small/*.lac were written by the reference itself (make_golden.py); the PCM they encode is
    regenerated from the recorded generator parameters."""
    with open(os.path.join(GOLDEN, "small", "index.json")) as f:
        index = json.load(f)
    for ent in index:
        g = ent["gen"]
        left, right = gpu.synth.synth_pcm(g["frames"], g["channels"], g["bit_depth"], g["sample_rate"], seed=g["seed"],
                                          kind=g["kind"], stereo=g["stereo"])
        with open(os.path.join(GOLDEN, "small", ent["name"] + ".lac"), "rb") as f:
            lac = f.read()
        assert hashlib.sha256(lac).hexdigest() == ent["lac_sha256"]
        l2, r2, info, _ = gpu.lacx.decode(lac)
        assert np.array_equal(l2, left), ent["name"]
        assert (right is None and r2 is None) or np.array_equal(r2, right), ent["name"]
        assert (info.sample_rate, info.bit_depth, info.channels, info.frames) == \
            (g["sample_rate"], g["bit_depth"], g["channels"], g["frames"])


CASES = [
    # frames, channels, bit_depth, rate, stereo_mode, kind, stereo, zero_run, partitioning
    (16384 * 5 + 37, 2, 16, 48000, 2, "music", "wide", True, True),
    (16384 * 4 + 4000, 2, 24, 96000, 2, "mixed", "wide", True, True),
    (16384 * 3, 2, 16, 44100, 2, "noise", "independent", True, True),
    (16384 * 2 + 1, 2, 24, 192000, 2, "music", "narrow", True, False),  # unpartitioned: the stateful adaptation
    (16384 * 3 + 100, 2, 16, 48000, 0, "mixed", "wide", True, False),
    (16384 * 3 + 100, 2, 16, 48000, 1, "mixed", "wide", False, True),
    (16384 * 3 + 5000, 1, 16, 48000, 0, "mixed", "wide", True, True),
    (2400, 1, 16, 48000, 0, "tone", "wide", True, True),
    (1, 2, 16, 48000, 2, "noise", "wide", True, True),
    (31, 2, 24, 48000, 2, "music", "identical", True, True),
    (257, 2, 16, 48000, 2, "walk", "wide", True, True),
    (4097, 2, 16, 48000, 2, "noise", "independent", True, True),
    (16384 + 300, 2, 16, 48000, 2, "silence", "identical", True, True),
    (16384 * 2, 2, 24, 48000, 2, "sparse", "half_silent", True, False),
    (16384 * 2 + 9, 2, 16, 96000, 2, "near_silence", "independent", True, False),
    (16384 * 2 + 9, 2, 16, 96000, 2, "near_silence", "independent", True, True),
    (16384 * 3, 2, 24, 48000, 2, "ramp", "wide", True, False),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(map(str, c)))
def test_round_trip_and_oracle_decoder(gpu, oracle, case):
    frames, ch, bd, sr, sm, kind, stereo, zr, pt = case
    left, right = gpu.synth.synth_pcm(frames, ch, bd, sr, seed=913, kind=kind, stereo=stereo)
    enc = gpu.lacx.Encoder(12, sm, sr, bd)
    enc.set_zero_run_enabled(zr)
    enc.set_partitioning_enabled(pt)
    lac = enc.encode(left, right)
    l2, r2, info, _ = gpu.lacx.decode(lac)
    assert np.array_equal(l2, left)
    assert (right is None and r2 is None) or np.array_equal(r2, right)
    lo, ro, hdr = oracle.decode(lac)
    assert np.array_equal(l2, lo) and (ro is None or np.array_equal(r2, ro))
    assert info.blocks == -(-frames // 16384) and info.stereo_mode == (sm if ch == 2 else 0)


def test_low_level_material_and_gaps(gpu):
    """Zero-run tokens, escapes, bin mode, long runs across partitions: scaled-down and gap-punched material."""
    rng = np.random.default_rng(12)
    for it in range(12):
        frames = int(rng.integers(20000, 70000))
        bd = int(rng.choice([16, 24]))
        kind = str(rng.choice(["music", "sparse", "walk", "noise", "near_silence"]))
        left, right = gpu.synth.synth_pcm(frames, 2, bd, 48000, seed=int(rng.integers(1, 10**6)), kind=kind, stereo="wide")
        sh = int(rng.integers(0, bd - 2))
        left, right = (left >> sh).astype(np.int32), (right >> sh).astype(np.int32)
        pos = 0
        while pos < frames:  # punch gaps of every run-length class into one channel
            pos += int(rng.choice([1, 2, 3, 7, 16, 40, 300]))
            gap = int(rng.choice([1, 3, 4, 5, 16, 64, 65, 1000, 20000]))
            left[pos:pos + gap] = 0
            pos += gap
        if it % 3 == 0:  # spikes: the 32-bit escape of the zero-run mode
            idx = rng.integers(0, frames, size=20)
            right[idx] = (1 << (bd - 1)) - 1
        enc = gpu.lacx.Encoder(12, 2, 48000, bd)
        enc.set_partitioning_enabled(bool(it & 1))
        l2, r2, _, _ = gpu.lacx.decode(enc.encode(left, right))
        assert np.array_equal(l2, left) and np.array_equal(r2, right), it


def test_full_size_round_trip(gpu):
    """BASELINE configs[1] size (10 min stereo 16/48, 1758 blocks): the device-made .lac matches the golden digest minted
    from the reference, and decodes back to the PCM it was made from."""
    with open(os.path.join(GOLDEN, "digests.json")) as f:
        ent = [e for e in json.load(f) if e["name"].startswith("cfg2")][0]
    g = ent["gen"]
    left, right = gpu.synth.synth_pcm(g["frames"], g["channels"], g["bit_depth"], g["sample_rate"], seed=g["seed"],
                                      kind=g["kind"], stereo=g["stereo"])
    lac = gpu.lacx.Encoder(12, ent["stereo_mode"], g["sample_rate"], g["bit_depth"]).encode(left, right)
    assert hashlib.sha256(lac).hexdigest() == ent["lac_sha256"]
    l2, r2, info, ms = gpu.lacx.decode(lac)
    assert info.blocks == 1758 and np.array_equal(l2, left) and np.array_equal(r2, right)
    print(f"decode kernels: {ms:.2f} ms for {2 * g['frames'] / 1e6:.1f} Msamples")


def test_damaged_streams_are_refused(gpu):
    left, right = gpu.synth.synth_pcm(16384 * 3 + 77, 2, 16, 48000, seed=5, kind="music")
    lac = gpu.lacx.Encoder(12, 2, 48000, 16).encode(left, right)
    with pytest.raises(RuntimeError, match="decode-error"):
        gpu.lacx.decode(lac[:-3])  # sizes no longer add up
    with pytest.raises(RuntimeError, match="decode-error"):
        gpu.lacx.decode(b"XX" + lac[2:])
    info = gpu.lacx.stream_parse(lac)
    head = 14 + 8 * info.blocks
    # cut the first block's payload short by moving bytes to the second block in the table: the first no longer ends
    # where its bitstream does
    bad = bytearray(lac)
    n0 = int.from_bytes(lac[18:22], "big")
    n1 = int.from_bytes(lac[26:30], "big")
    bad[18:22] = (n0 - 5).to_bytes(4, "big")
    bad[26:30] = (n1 + 5).to_bytes(4, "big")
    with pytest.raises(RuntimeError, match=r"decode-error\] block=0"):
        gpu.lacx.decode(bytes(bad))
    # an impossible predictor type in block 1's first channel header (byte 0 is the LR/MS flag)
    off1 = head + n0
    bad = bytearray(lac)
    bad[off1 + 1] = 7
    with pytest.raises(RuntimeError, match=r"decode-error\] block=1"):
        gpu.lacx.decode(bytes(bad))
    # the stream ends right behind a type-2 (LPC) channel header that announces 32 coefficients: the coefficient list
    # must be refused before it is read (it would reach 64 bytes past the payload)
    ml, _ = gpu.synth.synth_pcm(16384 + 100, 1, 16, 48000, seed=6, kind="music")
    mono = gpu.lacx.Encoder(12, 0, 48000, 16).encode(ml, None)
    mi = gpu.lacx.stream_parse(mono)
    assert mi.blocks == 2
    mhead = 14 + 8 * mi.blocks
    m0 = int.from_bytes(mono[18:22], "big")
    cut = bytearray(mono[:mhead + m0] + b"\x02\x20")
    cut[26:30] = (2).to_bytes(4, "big")
    with pytest.raises(RuntimeError, match=r"decode-error\] block=1"):
        gpu.lacx.decode(bytes(cut))
    # random damage inside the payload: refused or decoded to something else, never a hang or a crash
    rng = np.random.default_rng(3)
    for _ in range(20):
        bad = bytearray(lac)
        for pos in rng.integers(head, len(lac), size=3):
            bad[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            l2, r2, _, _ = gpu.lacx.decode(bytes(bad))
            assert l2.size == left.size
        except RuntimeError as err:
            assert "decode-error" in str(err)


class _Bits:
    """MSB-first bit writer (the layout of ref src/codec/bitstream/bit_writer.cpp), for hand-made channel blocks."""

    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, value, bits):
        assert 0 <= value < (1 << bits) or bits == 0
        self.v = (self.v << bits) | value
        self.n += bits

    def rice(self, u, k):
        self.put((1 << (u >> k)) - 1, u >> k)  # unary quotient: ones ...
        self.put(0, 1)                         # ... and their terminator
        self.put(u & ((1 << k) - 1), k)

    def bytes(self):
        pad = (-self.n) % 8
        return ((self.v << pad)).to_bytes((self.n + pad) // 8, "big")


def _zigzag(x):
    return (x << 1) ^ (x >> 63)


def test_foreign_streams_high_lpc_orders_and_escapes(gpu, oracle):
    """Streams no test encoder writes but the format allows (ref block/decoder.cpp:64-520): LPC orders 13..32 (taps beyond
    the register window come from the history ring in LDS), a static-Rice partition, and zero-run mode's 32-bit escape
    with large values -- channel blocks made by hand with a bit writer, the device decoder against the oracle's (itself
    pinned against the reference's)."""
    rng = np.random.default_rng(8)
    n = 1500
    for order in (13, 14, 31, 32):
        coefs = [int(c) for c in rng.integers(-900, 901, size=order)]  # sum |c| < 2^15: the synthesis stays small
        res = [int(r) for r in rng.integers(-300, 301, size=n)]
        k = 6
        w = _Bits()
        w.put(2, 8)               # type: LPC
        w.put(order, 8)
        for c in coefs:
            w.put(c & 0xFFFF, 16)
        w.put((3 << 5) | 0, 8)    # control: no partitioning, mode of the block = static Rice
        w.put(3, 2)               # partition table, one entry: mode, k
        w.put(k, 5)
        for r in res:
            w.rice(_zigzag(r), k)
        pay = w.bytes()
        lac = gpu.lacx.assemble(48000, 24, 0, 1, [(pay, np.array([[n, len(pay)]], dtype=np.uint32))])
        lo, ro, _ = oracle.decode(lac)
        l2, r2, info, _ = gpu.lacx.decode(lac)
        assert ro is None and r2 is None and info.frames == n
        assert np.array_equal(l2, lo), order
        assert np.abs(lo).max() > 300  # the predictor did something
    # zero-run mode, every sample as the 32-bit escape (tag 2 + 32 bits), values up to the 24-bit range
    vals = [int(v) for v in rng.integers(-(1 << 23), 1 << 23, size=n)]
    vals[:4] = [(1 << 23) - 1, -(1 << 23), 0, 1]

    def escape_block(values):
        w = _Bits()
        w.put(0, 8)               # type: fixed predictor ...
        w.put(0, 8)               # ... of order 0: the residual is the sample
        w.put((1 << 5) | 0, 8)    # control: no partitioning, zero-run mode
        w.put(1, 2)
        w.put(9, 5)
        for v in values:
            w.put(2, 2)
            w.put(_zigzag(v) & 0xFFFFFFFF, 32)
        return w.bytes()

    pay = escape_block(vals)
    lac = gpu.lacx.assemble(48000, 24, 0, 1, [(pay, np.array([[n, len(pay)]], dtype=np.uint32))])
    lo, _, _ = oracle.decode(lac)
    l2, _, _, _ = gpu.lacx.decode(lac)
    assert np.array_equal(l2, lo) and np.array_equal(l2, np.array(vals, dtype=np.int32))
    # a magnitude the encoder's domain cannot produce (>= 2^30) is refused, not decoded differently from the reference
    pay = escape_block(vals[:10] + [1 << 29] + vals[11:])
    lac = gpu.lacx.assemble(48000, 24, 0, 1, [(pay, np.array([[n, len(pay)]], dtype=np.uint32))])
    with pytest.raises(RuntimeError, match=r"decode-error\] block=0"):
        gpu.lacx.decode(lac)


def test_legacy_version_2_container(gpu):
    """Version 2 of the container has no compressed block sizes (ref lac/decoder.cpp:100-104, 209-219): the same block
    payloads back to back behind a table of frame counts.  Rebuilt here from a version-3 stream; one lane walks it."""
    for frames, ch, bd, sm in ((16384 * 3 + 500, 2, 16, 2), (16384 * 2 + 7, 1, 24, 0), (300, 2, 16, 1)):
        left, right = gpu.synth.synth_pcm(frames, ch, bd, 48000, seed=77, kind="mixed")
        v3 = gpu.lacx.Encoder(12, sm, 48000, bd).encode(left, right)
        info = gpu.lacx.stream_parse(v3)
        nb = info.blocks
        table = b"".join(v3[14 + 8 * b:18 + 8 * b] for b in range(nb))
        v2 = v3[:2] + bytes([2]) + v3[3:14] + table + v3[14 + 8 * nb:]
        i2 = gpu.lacx.stream_parse(v2)
        assert i2 is not None and i2.version == 2 and i2.frames == frames and i2.blocks == nb
        l2, r2, _, _ = gpu.lacx.decode(v2)
        assert np.array_equal(l2, left) and (right is None or np.array_equal(r2, right))
        with pytest.raises(RuntimeError, match="decode-error"):
            gpu.lacx.decode(v2 + b"\0")  # trailing frame payload
        with pytest.raises(RuntimeError, match="decode-error"):
            gpu.lacx.decode(v2[:-1])


def test_decoder_handle_reuses_its_buffers(gpu, oracle):
    """lacx_decoder_create / lacx_decoder_decode: one handle through streams of different sizes, formats and channel counts
    (buffers grow, never shrink), a malformed stream in between, and the handle-less lacx_decode beside it."""
    dec = gpu.lacx.Decoder(device=0, reuse_output=True)
    for frames, ch, bd, sr, sm, kind in ((16384 * 3 + 9, 2, 16, 48000, 2, "mixed"), (16384 * 9 + 100, 2, 24, 96000, 2, "music"),
                                          (5000, 1, 16, 44100, 0, "noise"), (16384 * 9 + 100, 2, 24, 96000, 0, "sparse"),
                                          (16384 * 20, 1, 24, 192000, 0, "music")):
        left, right = gpu.synth.synth_pcm(frames, ch, bd, sr, seed=90, kind=kind)
        lac = oracle.encode(left, right, sr, bd, sm, threads=8)
        for _ in range(2):
            l, r, info, ms = dec.decode(lac)
            assert np.array_equal(l, left) and (r is None) == (ch == 1) and (ch == 1 or np.array_equal(r, right))
            assert (info.frames, info.channels, info.bit_depth) == (frames, ch, bd) and ms > 0
        bad = bytearray(lac)
        bad[len(bad) // 2] ^= 0x55
        try:
            dec.decode(bytes(bad))
        except RuntimeError as ex:
            assert "[decode-error]" in str(ex)
        l2, r2, _, _ = gpu.lacx.decode(lac)
        assert np.array_equal(l2, left)
    dec.close()
