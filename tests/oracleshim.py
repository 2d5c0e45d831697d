"""ctypes view of oracle/liblac_oracle.so (plain-C restatement; test infrastructure only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liblac_oracle.so")

MAX_PARTS = 256


class Plan(C.Structure):
    _fields_ = [
        ("predictor_type", C.c_uint8),
        ("order", C.c_uint8),
        ("partition_order", C.c_uint8),
        ("reserved", C.c_uint8),
        ("coeffs_q15", C.c_int16 * 13),
        ("reserved2", C.c_uint16),
        ("part_count", C.c_uint32),
        ("total_bits", C.c_uint64),
        ("best_bits", C.c_uint64),
        ("part_mode", C.c_uint8 * MAX_PARTS),
        ("part_k", C.c_uint8 * MAX_PARTS),
    ]


class Stereo(C.Structure):
    _fields_ = [("choose_ms", C.c_int), ("uncertain", C.c_int), ("sums", C.c_uint64 * 12)]


def build():
    src = os.path.join(ORACLE_DIR, "lac_oracle.c")
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liblac_oracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(ORACLE_SO)
        _lib.laco_free.argtypes = [C.c_void_p]
    return _lib


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def encode(left, right=None, sample_rate=48000, bit_depth=16, stereo_mode=2, zero_run=True,
           partitioning=True, threads=1) -> bytes:
    L, lp = _i32(left)
    rp = None
    if right is not None:
        R, rp = _i32(right)
    out = C.POINTER(C.c_uint8)()
    size = C.c_uint64()
    rc = lib().laco_encode(lp, rp, C.c_uint64(L.size), C.c_uint32(sample_rate), bit_depth, stereo_mode,
                           int(zero_run), int(partitioning), threads, C.byref(out), C.byref(size))
    if rc == 1:
        raise ValueError("invalid argument")
    if rc != 0:
        raise RuntimeError("encode failed")
    data = C.string_at(out, size.value)
    lib().laco_free(out)
    return data


def block_encode(pcm, zero_run=True, partitioning=True) -> bytes:
    P, pp = _i32(pcm)
    out = C.POINTER(C.c_uint8)()
    size = C.c_uint64()
    lib().laco_block_encode(pp, C.c_uint32(P.size), int(zero_run), int(partitioning), C.byref(out),
                            C.byref(size))
    data = C.string_at(out, size.value)
    lib().laco_free(out)
    return data


def block_plan(pcm, zero_run=True, partitioning=True) -> Plan:
    P, pp = _i32(pcm)
    plan = Plan()
    lib().laco_block_plan(pp, C.c_uint32(P.size), int(zero_run), int(partitioning), C.byref(plan))
    return plan


def autocorr(pcm, order=12):
    P, pp = _i32(pcm)
    r = np.zeros(order + 1, dtype=np.int64)
    lib().laco_autocorr(pp, C.c_uint32(P.size), order, r.ctypes.data_as(C.POINTER(C.c_int64)))
    return r


def lpc_analyze(pcm, order):
    P, pp = _i32(pcm)
    co = np.zeros(order + 1, dtype=np.int16)
    lib().laco_lpc_analyze.restype = C.c_int
    used = lib().laco_lpc_analyze(pp, C.c_uint32(P.size), order, co.ctypes.data_as(C.POINTER(C.c_int16)))
    return used, co


def levinson_q15(r, order):
    R = np.ascontiguousarray(r, dtype=np.int64)
    co = np.zeros(order + 1, dtype=np.int16)
    lib().laco_levinson_q15.restype = C.c_int
    used = lib().laco_levinson_q15(R.ctypes.data_as(C.POINTER(C.c_int64)), order,
                                   co.ctypes.data_as(C.POINTER(C.c_int16)))
    return used, co


def stereo_estimate(left, right) -> Stereo:
    L, lp = _i32(left)
    R, rp = _i32(right)
    s = Stereo()
    lib().laco_stereo_estimate(lp, rp, C.c_uint32(L.size), C.byref(s))
    return s


def adapt_k_sequence(u):
    U = np.ascontiguousarray(u, dtype=np.uint32)
    out = np.zeros(U.size, dtype=np.uint32)
    lib().laco_adapt_k_sequence(U.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(U.size),
                                out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def decode(data: bytes):
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    lp = C.POINTER(C.c_int32)()
    rp = C.POINTER(C.c_int32)()
    frames = C.c_uint64()
    ch = C.c_int()
    sr = C.c_uint32()
    bd = C.c_int()
    sm = C.c_int()
    rc = lib().laco_decode(buf, C.c_uint64(len(data)), C.byref(lp), C.byref(rp), C.byref(frames),
                           C.byref(ch), C.byref(sr), C.byref(bd), C.byref(sm))
    if rc != 0:
        raise RuntimeError("decode failed")
    n = frames.value
    left = np.ctypeslib.as_array(lp, shape=(n,)).copy() if n else np.zeros(0, np.int32)
    lib().laco_free(lp)
    right = None
    if ch.value == 2:
        right = np.ctypeslib.as_array(rp, shape=(n,)).copy() if n else np.zeros(0, np.int32)
        lib().laco_free(rp)
    return left, right, dict(channels=ch.value, sample_rate=sr.value, bit_depth=bd.value,
                             stereo_mode=sm.value)
