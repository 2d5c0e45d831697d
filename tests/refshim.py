"""ctypes view of oracle/_ref/liblac_ref.so (the unmodified reference, test infrastructure only)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "liblac_ref.so")


def available() -> bool:
    return os.path.exists(REF_SO)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(REF_SO)
        _lib.lacref_last_error.restype = C.c_char_p
        _lib.lacref_free.argtypes = [C.c_void_p]
        _lib.lacref_encode.restype = C.c_int
        _lib.lacref_decode.restype = C.c_int
        _lib.lacref_block_encode.restype = C.c_int
        _lib.lacref_lpc_analyze.restype = C.c_int
    return _lib


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def encode(left, right=None, sample_rate=48000, bit_depth=16, stereo_mode=2, zero_run=True,
           partitioning=True, threads=0) -> bytes:
    L, lp = _i32(left)
    if right is not None:
        R, rp = _i32(right)
    else:
        rp = None
    out = C.POINTER(C.c_uint8)()
    size = C.c_uint64()
    rc = lib().lacref_encode(lp, rp, C.c_uint64(L.size), 2 if right is not None else 1,
                             C.c_uint32(sample_rate), bit_depth, stereo_mode, int(zero_run),
                             int(partitioning), threads, C.byref(out), C.byref(size))
    if rc != 0:
        msg = lib().lacref_last_error().decode()
        raise {1: ValueError, 2: RuntimeError}.get(rc, RuntimeError)(msg)
    data = C.string_at(out, size.value)
    lib().lacref_free(out)
    return data


def decode(data: bytes):
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    lp = C.POINTER(C.c_int32)()
    rp = C.POINTER(C.c_int32)()
    frames = C.c_uint64()
    ch = C.c_int()
    sr = C.c_uint32()
    bd = C.c_int()
    sm = C.c_int()
    rc = lib().lacref_decode(buf, C.c_uint64(len(data)), C.byref(lp), C.byref(rp), C.byref(frames),
                             C.byref(ch), C.byref(sr), C.byref(bd), C.byref(sm))
    if rc != 0:
        raise RuntimeError(lib().lacref_last_error().decode())
    n = frames.value
    left = np.ctypeslib.as_array(lp, shape=(n,)).copy()
    lib().lacref_free(lp)
    right = None
    if ch.value == 2:
        right = np.ctypeslib.as_array(rp, shape=(n,)).copy()
        lib().lacref_free(rp)
    return left, right, dict(channels=ch.value, sample_rate=sr.value, bit_depth=bd.value,
                             stereo_mode=sm.value)


def block_encode(pcm, zero_run=True, partitioning=True) -> bytes:
    P, pp = _i32(pcm)
    out = C.POINTER(C.c_uint8)()
    size = C.c_uint64()
    rc = lib().lacref_block_encode(pp, C.c_uint32(P.size), int(zero_run), int(partitioning),
                                   C.byref(out), C.byref(size))
    if rc != 0:
        raise RuntimeError(lib().lacref_last_error().decode())
    data = C.string_at(out, size.value)
    lib().lacref_free(out)
    return data


def lpc_analyze(pcm, order):
    P, pp = _i32(pcm)
    co = np.zeros(order + 1, dtype=np.int16)
    used = lib().lacref_lpc_analyze(pp, C.c_uint32(P.size), order,
                                    co.ctypes.data_as(C.POINTER(C.c_int16)))
    return used, co


def adapt_k_sequence(u):
    U = np.ascontiguousarray(u, dtype=np.uint32)
    out = np.zeros(U.size, dtype=np.uint32)
    lib().lacref_adapt_k_sequence(U.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(U.size),
                                  out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def read_wav(path: str):
    """Reference read_wav: None when rejected, else (left, right_or_None, channels, sample_rate, bit_depth)."""
    ch, sr, bd, fr = C.c_uint16(), C.c_uint32(), C.c_uint8(), C.c_uint64()
    cap = max(1, os.path.getsize(path))
    left = np.zeros(cap, dtype=np.int32)
    right = np.zeros(cap, dtype=np.int32)
    ok = lib().lacref_read_wav(path.encode(), C.byref(ch), C.byref(sr), C.byref(bd), C.byref(fr),
                               left.ctypes.data_as(C.POINTER(C.c_int32)), right.ctypes.data_as(C.POINTER(C.c_int32)),
                               C.c_uint64(cap))
    if not ok:
        return None
    n = fr.value
    return left[:n].copy(), (right[:n].copy() if ch.value == 2 else None), ch.value, sr.value, bd.value
