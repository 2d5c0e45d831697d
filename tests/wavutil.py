"""WAV file images for the ingest tests: a canonical writer plus knobs that produce the
malformed variants the reference's read_wav rejects (ref src/io/wav_io.cpp:167-277)."""
import struct

import numpy as np


def pcm_bytes(left, right, bit_depth):
    ch = 1 if right is None else 2
    if ch == 2:
        inter = np.empty(left.size * 2, dtype=np.int32)
        inter[0::2] = left
        inter[1::2] = right
    else:
        inter = np.asarray(left, dtype=np.int32)
    if bit_depth == 16:
        return inter.astype("<i2").tobytes()
    u = inter.astype(np.uint32)
    out = np.empty((inter.size, 3), dtype=np.uint8)
    out[:, 0] = u & 0xFF
    out[:, 1] = (u >> 8) & 0xFF
    out[:, 2] = (u >> 16) & 0xFF
    return out.tobytes()


def chunk(cid: bytes, body: bytes, pad=True, size=None):
    n = len(body) if size is None else size
    return cid + struct.pack("<I", n) + body + (b"\0" if pad and (len(body) & 1) else b"")


def fmt_chunk(channels, rate, bits, fmt=1, align=None, byte_rate=None, size=16, extra=b""):
    a = channels * (bits // 8) if align is None else align
    br = rate * a if byte_rate is None else byte_rate
    return chunk(b"fmt ", struct.pack("<HHIIHH", fmt, channels, rate, br, a, bits) + extra, size=size)


def riff(chunks, riff_size=None, form=b"WAVE", tag=b"RIFF"):
    body = form + b"".join(chunks)
    return tag + struct.pack("<I", len(body) if riff_size is None else riff_size) + body


def make_wav(left, right, rate, bits, before=(), between=(), after=()):
    ch = 1 if right is None else 2
    return riff(list(before) + [fmt_chunk(ch, rate, bits)] + list(between) + [chunk(b"data", pcm_bytes(left, right, bits))]
                + list(after))
