"""Host half of the interleaved WAV ingest: lacx_wav_parse accepts / rejects exactly what the
reference's read_wav does (ref src/io/wav_io.cpp:167-277) and points at the raw data chunk.  The expectations
are written out here; where the reference build is present (build container) every case is also run through
the real read_wav."""
import os
import tempfile

import numpy as np
import pytest

import refshim
import wavutil as W


def _cases(synth):
    l16, r16 = synth.synth_pcm(1000, 2, 16, 48000, seed=1, kind="music")
    l24, r24 = synth.synth_pcm(777, 2, 24, 96000, seed=2, kind="mixed")
    d16 = W.chunk(b"data", W.pcm_bytes(l16, r16, 16))
    f16 = W.fmt_chunk(2, 48000, 16)
    odd = W.chunk(b"LIST", b"abc")          # odd size: one pad byte follows
    cases = [
        ("stereo16", W.make_wav(l16, r16, 48000, 16), True),
        ("mono16", W.make_wav(l16, None, 44100, 16), True),
        ("stereo24", W.make_wav(l24, r24, 96000, 24), True),
        ("mono24_192k", W.make_wav(l24, None, 192000, 24), True),
        ("extra_chunks", W.make_wav(l16, r16, 48000, 16, before=[odd], between=[W.chunk(b"fact", b"1234")],
                                    after=[odd, W.chunk(b"id3 ", b"")]), True),
        ("one_frame", W.make_wav(l16[:1], r16[:1], 48000, 16), True),
        ("odd_data_mono24", W.make_wav(l24[:5], None, 48000, 24), True),   # 15 data bytes + pad
        ("too_short", b"RIFF\x04\0\0\0WAV", False),
        ("bad_magic", W.riff([f16, d16], tag=b"RIFX"), False),
        ("bad_form", W.riff([f16, d16], form=b"AVI "), False),
        ("riff_size_small", W.riff([f16, d16], riff_size=10), False),
        ("riff_size_big", W.riff([f16, d16]) + b"\0\0", False),
        ("no_data", W.riff([f16]), False),
        ("no_fmt", W.riff([d16]), False),
        ("data_before_fmt", W.riff([d16, f16]), False),
        ("two_fmt", W.riff([f16, f16, d16]), False),
        ("two_data", W.riff([f16, d16, d16]), False),
        ("fmt_18", W.riff([W.fmt_chunk(2, 48000, 16, size=18, extra=b"\0\0"), d16]), False),
        ("float_format", W.riff([W.fmt_chunk(2, 48000, 16, fmt=3), d16]), False),
        ("extensible", W.riff([W.fmt_chunk(2, 48000, 16, fmt=0xFFFE), d16]), False),
        ("bits_8", W.riff([W.fmt_chunk(2, 48000, 8), d16]), False),
        ("bits_32", W.riff([W.fmt_chunk(2, 48000, 32), d16]), False),
        ("rate_22050", W.riff([W.fmt_chunk(2, 22050, 16), d16]), False),
        ("channels_3", W.riff([W.fmt_chunk(3, 48000, 16), d16]), False),
        ("channels_0", W.riff([W.fmt_chunk(0, 48000, 16), d16]), False),
        ("bad_align", W.riff([W.fmt_chunk(2, 48000, 16, align=2), d16]), False),
        ("bad_byte_rate", W.riff([W.fmt_chunk(2, 48000, 16, byte_rate=1), d16]), False),
        ("empty_data", W.riff([f16, W.chunk(b"data", b"")]), False),
        ("partial_frame", W.riff([f16, W.chunk(b"data", b"\0" * 6)]), False),
        ("chunk_overruns", W.riff([f16, W.chunk(b"data", b"\0" * 8, size=400)]), False),
        ("trailing_garbage_7", W.riff([f16, d16, b"junk\0\0\0"]), False),
        ("missing_pad", W.riff([f16, W.chunk(b"LIST", b"abc", pad=False), d16]), False),
    ]
    return cases, {"stereo16": (l16, r16), "mono16": (l16, None), "stereo24": (l24, r24), "mono24_192k": (l24, None),
                   "extra_chunks": (l16, r16), "one_frame": (l16[:1], r16[:1]), "odd_data_mono24": (l24[:5], None)}


def _decode(wav, info):
    raw = np.frombuffer(wav, dtype=np.uint8, count=info.data_bytes, offset=info.data_offset)
    if info.bit_depth == 16:
        s = raw.view("<i2").astype(np.int32)
    else:
        b = raw.reshape(-1, 3).astype(np.uint32)
        s = (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)).astype(np.uint32)
        s = ((s << 8).astype(np.int32) >> 8)
    return (s[0::2], s[1::2]) if info.channels == 2 else (s, None)


def test_wav_parse_accepts_and_rejects_like_read_wav(pkg):
    cases, pcm = _cases(pkg.synth)
    for name, wav, ok in cases:
        info = pkg.lacx.wav_parse(wav)
        assert (info is not None) == ok, name
        if ok:
            left, right = _decode(wav, info)
            assert info.frames == pcm[name][0].size, name
            assert np.array_equal(left, pcm[name][0]), name
            assert (right is None) == (pcm[name][1] is None) and (right is None or np.array_equal(right, pcm[name][1])), name
    assert pkg.lacx.wav_parse(b"") is None


@pytest.mark.skipif(not refshim.available(), reason="reference build (oracle/_ref) not present")
def test_wav_parse_cross_checked_with_the_reference_reader(pkg):
    cases, _ = _cases(pkg.synth)
    with tempfile.TemporaryDirectory() as d:
        for name, wav, ok in cases:
            path = os.path.join(d, name + ".wav")
            with open(path, "wb") as f:
                f.write(wav)
            ref = refshim.read_wav(path)
            assert (ref is not None) == ok, f"expectation for {name} disagrees with the reference"
            info = pkg.lacx.wav_parse(wav)
            assert (info is not None) == (ref is not None), name
            if ref is not None:
                rl, rr, ch, sr, bd = ref
                assert (info.channels, info.sample_rate, info.bit_depth, info.frames) == (ch, sr, bd, rl.size), name
                left, right = _decode(wav, info)
                assert np.array_equal(left, rl) and (rr is None or np.array_equal(right, rr)), name
