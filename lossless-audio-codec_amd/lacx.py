"""ctypes binding of liblacx.so (include/lacx.h) and Python mirrors of the reference's encoder classes.

`Encoder` mirrors LAC::Encoder (ref src/codec/lac/encoder.hpp:12-43): same constructor argument
order, the same setters, `encode(left, right)` returning the .lac bytes, ValueError where the
reference throws std::invalid_argument and RuntimeError where it throws std::runtime_error.
`BlockEncoder` mirrors Block::Encoder (ref src/codec/block/encoder.hpp:9-30).

PyTorch is optional plumbing: `encode_tensors` takes int32 CUDA(HIP) tensors whose storage is handed
to the library by raw pointer (no torch types cross the ABI).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liblacx.so")
# the diagnostic twin (analysis kernel built with -DLACX_TEST_HOOKS: honours LACX_DEBUG_SKIP); tests and timing
# experiments select it explicitly with use_library(HOOKS_LIB_PATH) -- the product never loads it by itself
HOOKS_LIB_PATH = os.path.join(HERE, "liblacx_hooks.so")
DEVICE_ALL = -2
MAX_FANOUT = 16

OK, E_INVALID, E_RUNTIME, E_DEVICE = 0, 1, 2, 3
MAX_BLOCK = 16384
SLOTS = 16
CH_L, CH_R, CH_M, CH_S = 0, 1, 2, 3


class Config(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_uint32),
        ("bit_depth", C.c_uint8),
        ("stereo_mode", C.c_uint8),
        ("zero_run_enabled", C.c_uint8),
        ("partitioning_enabled", C.c_uint8),
        ("device", C.c_int32),
        ("emit_threads", C.c_uint32),
        ("flags", C.c_uint32),
    ]


class ChannelPlan(C.Structure):
    _fields_ = [
        ("predictor_type", C.c_uint8),
        ("order", C.c_uint8),
        ("partition_order", C.c_uint8),
        ("valid", C.c_uint8),
        ("coef", C.c_int16 * 12),
        ("payload_bytes", C.c_uint32),
        ("total_bits", C.c_uint64),
        ("part_mode_k", C.c_uint8 * 256),
    ]


class BlockPlan(C.Structure):
    _fields_ = [
        ("choose_ms", C.c_uint8),
        ("uncertain", C.c_uint8),
        ("est_ms", C.c_uint8),
        ("invalid", C.c_uint8),
        ("frames", C.c_uint32),
        ("first_bad", C.c_uint32),
        ("pad", C.c_uint32),
    ]


class Timing(C.Structure):
    _fields_ = [
        ("h2d_ms", C.c_double),
        ("analysis_ms", C.c_double),
        ("ingest_ms", C.c_double),
        ("probe_ms", C.c_double),
        ("full_ms", C.c_double),
        ("d2h_ms", C.c_double),
        ("emit_ms", C.c_double),
        ("total_ms", C.c_double),
        ("full_slots", C.c_uint64),
        ("probe_slots", C.c_uint64),
        ("full_launches", C.c_uint32),
        ("regrows", C.c_uint32),
        ("full_exec_ms", C.c_double),
        ("emit_direct", C.c_uint32),
        ("moved_by_k_pack", C.c_uint32),
        ("packer_gave_up", C.c_uint32),
        ("drain_copies", C.c_uint32),
        ("drain_first_ms", C.c_double),
        ("drain_last_ms", C.c_double),
        ("poll_gap_max_ms", C.c_double),
        ("kernels_done_ms", C.c_double),
        ("enqueue_ms", C.c_double),
        ("silent_copies", C.c_uint32),
        ("reserved0", C.c_uint32),
    ]


EXPORTS = (
    "lacx_encoder_create", "lacx_encoder_destroy", "lacx_last_error", "lacx_free", "lacx_get_timing",
    "lacx_encode", "lacx_encode_device", "lacx_analyze", "lacx_analyze_device", "lacx_emit_from_plans",
    "lacx_encode_shard", "lacx_encode_shard_device", "lacx_encode_shard_device_view", "lacx_encode_shard_pcm_device_view", "lacx_assemble", "lacx_block_encode",
    "lacx_block_plan_only", "lacx_debug_lpc", "lacx_debug_stamps", "lacx_device_count", "lacx_wav_parse",
    "lacx_encode_wav", "lacx_encode_shard_pcm_device_begin", "lacx_encode_shard_end", "lacx_debug_emit_workers", "lacx_encode_wav_view",
    "lacx_stream_parse", "lacx_decode", "lacx_decode_last_error", "lacx_encode_batch_device",
    "lacx_encoder_create_multi", "lacx_encoder_lanes", "lacx_fanout_range", "lacx_encode_fanout_resident",
    "lacx_get_fanout_stats", "lacx_get_lane_timing", "lacx_fanout_exchange_note",
    "lacx_decoder_create", "lacx_decoder_destroy", "lacx_decoder_decode", "lacx_sizeof",
)


def build(force: bool = False) -> str:
    """Compiles liblacx.so in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", HERE, "liblacx.so", "liblacx_hooks.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None
_lib_path = LIB_PATH
_libs = {}


def use_library(path: str | None = None):
    """Selects the shared library the binding calls from now on (None: the product's liblacx.so).  For the parity tests
    that need the hooks build and for A/B experiments of differently built libraries (scripts/kexp.py); encoders created
    before the switch must not be used after it."""
    global _lib, _lib_path
    _lib_path = path or LIB_PATH
    _lib = None


def lib():
    global _lib
    if _lib is None:
        path = _lib_path
        if path in _libs:
            _lib = _libs[path]
            return _lib
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `make -C {HERE}` (or __graft_entry__.build()); "
                "the LAC encode path has no Python/CPU fallback")
        L = C.CDLL(path)
        L.lacx_last_error.restype = C.c_char_p
        L.lacx_last_error.argtypes = [C.c_void_p]
        L.lacx_free.argtypes = [C.c_void_p]
        L.lacx_encoder_destroy.argtypes = [C.c_void_p]
        L.lacx_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
        L.lacx_device_count.restype = C.c_int
        L.lacx_decode_last_error.restype = C.c_char_p
        L.lacx_fanout_exchange_note.restype = C.c_char_p
        L.lacx_fanout_exchange_note.argtypes = [C.c_void_p]
        L.lacx_encoder_lanes.restype = C.c_uint32
        L.lacx_encoder_lanes.argtypes = [C.c_void_p]
        L.lacx_fanout_range.restype = None
        # the structs declared in this file against the library's own sizeof(): a layout that has drifted from
        # include/lacx.h would otherwise show up as memory corruption behind the first call that fills one
        L.lacx_sizeof.restype = C.c_uint32
        L.lacx_sizeof.argtypes = [C.c_char_p]
        for name, cls in abi_structs().items():
            want = int(L.lacx_sizeof(name.encode()))
            if want != C.sizeof(cls):
                raise RuntimeError(f"{path}: sizeof(lacx_{name}) is {want}, the Python binding declares {C.sizeof(cls)} bytes")
        _libs[path] = L
        _lib = L
    return _lib


def abi_structs() -> dict:
    """include/lacx.h struct name (without the prefix) -> the ctypes class that mirrors it."""
    return {"config": Config, "channel_plan": ChannelPlan, "block_plan": BlockPlan, "timing": Timing, "pcm": Pcm,
            "batch_item": BatchItem, "batch_out": BatchOut, "wav_info": WavInfo, "fanout_shard": FanoutShard,
            "fanout_out": FanoutOut, "fanout_stats": FanoutStats, "stream_info": StreamInfo}


def device_count() -> int:
    return int(lib().lacx_device_count())


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def _raise(handle, rc):
    msg = lib().lacx_last_error(handle).decode(errors="replace")
    if rc == E_INVALID:
        raise ValueError(msg)
    raise RuntimeError(msg)


def _take(ptr, size) -> bytes:
    data = C.string_at(ptr, size.value)
    lib().lacx_free(ptr)
    return data


class Encoder:
    """Mirror of LAC::Encoder (ref src/codec/lac/encoder.hpp:12-43)."""

    def __init__(self, order: int = 12, stereo_mode: int = 0, sample_rate: int = 44100, bit_depth: int = 16,
                 debug_lpc: bool = False, debug_stereo_est: bool = False, debug_zr: bool = False,
                 device: int = -1, devices=None, min_blocks_per_device: int = 0):
        """device: HIP ordinal, -1 = the current device, DEVICE_ALL = every visible device.  devices: an explicit device list
        (lacx_encoder_create_multi): encode / encode_wav / encode_wav_view spread the stream's blocks over them."""
        self.order = order  # stored and ignored, as in the reference (ref block/encoder.cpp:41)
        self._cfg = Config(sample_rate & 0xFFFFFFFF, bit_depth & 0xFF, stereo_mode & 0xFF, 1, 1, device, 0, 0)
        self._raw_stereo_mode = stereo_mode
        self._devices = None if devices is None else [int(d) for d in devices]
        self._min_blocks = int(min_blocks_per_device)
        self._h = None

    # -- setters of the reference ------------------------------------------------------------
    def set_zero_run_enabled(self, enabled: bool):
        self._cfg.zero_run_enabled = 1 if enabled else 0
        self._reset()

    def set_partitioning_enabled(self, enabled: bool):
        self._cfg.partitioning_enabled = 1 if enabled else 0
        self._reset()

    def set_debug_partitions(self, enabled: bool):  # debug prints only in the reference
        pass

    def set_thread_count(self, max_threads: int):
        self._cfg.emit_threads = int(max_threads)
        self._reset()

    def set_host_emit(self, enabled: bool):
        """True keeps the bit emit on the host (north_star layout); default is the device-side emit."""
        self._cfg.flags = (self._cfg.flags | 1) if enabled else (self._cfg.flags & ~1)
        self._reset()

    # -- plumbing ------------------------------------------------------------------------------
    def _reset(self):
        if self._h is not None:
            lib().lacx_encoder_destroy(self._h)
            self._h = None

    def _handle(self):
        if self._h is None:
            h = C.c_void_p()
            if self._devices is not None:
                devs = (C.c_int32 * len(self._devices))(*self._devices)
                rc = lib().lacx_encoder_create_multi(C.byref(self._cfg), devs, C.c_uint32(len(self._devices)),
                                                     C.c_uint32(self._min_blocks), C.byref(h))
            else:
                rc = lib().lacx_encoder_create(C.byref(self._cfg), C.byref(h))
            if rc != OK:
                raise RuntimeError("lacx_encoder_create failed")
            self._h = h
        return self._h

    # -- the fan-out over several devices ------------------------------------------------------------
    def lanes(self) -> int:
        return int(lib().lacx_encoder_lanes(self._handle()))

    def fanout_stats(self) -> "FanoutStats":
        st = FanoutStats()
        lib().lacx_get_fanout_stats(self._handle(), C.byref(st))
        return st

    def fanout_exchange_note(self) -> str:
        return lib().lacx_fanout_exchange_note(self._handle()).decode(errors="replace")

    def lane_timing(self, lane: int) -> Timing:
        t = Timing()
        if lib().lacx_get_lane_timing(self._handle(), C.c_uint32(lane), C.byref(t)) != OK:
            raise ValueError("no such lane")
        return t

    def encode_fanout_resident(self, shards):
        """shards: [(data_ptr, layout, channels, frames[, data1_ptr]), ...], shard g resident on the device of lane g.
        Returns [(PayloadView, table, device, byte_offset), ...] (views into the lanes' pinned result regions)."""
        n = len(shards)
        ins = (FanoutShard * n)()
        for it, sh in zip(ins, shards):
            it.pcm = Pcm(sh[0], sh[4] if len(sh) > 4 else None, sh[1], sh[2])
            it.frames = sh[3]
        outs = (FanoutOut * n)()
        h = self._handle()
        rc = lib().lacx_encode_fanout_resident(h, ins, C.c_uint32(n), outs)
        if rc != OK:
            _raise(h, rc)
        return [(PayloadView(o.payload, o.payload_size), np.ctypeslib.as_array(o.table, shape=(o.nblocks, 2)), int(o.device),
                 int(o.byte_offset)) for o in outs]

    def close(self):
        self._reset()

    def __del__(self):
        try:
            self._reset()
        except Exception:
            pass

    def timing(self) -> Timing:
        t = Timing()
        lib().lacx_get_timing(self._handle(), C.byref(t))
        return t

    # -- LAC::Encoder::encode ------------------------------------------------------------------
    def encode(self, left, right=None) -> bytes:
        L, lp = _i32(left)
        rp = None
        if right is not None and len(right) != 0:
            R, rp = _i32(right)
            if R.size != L.size:
                raise ValueError(f"right channel size ({R.size}) must match left channel size ({L.size})")
        if L.size == 0:
            raise ValueError("left channel must not be empty")
        out = C.POINTER(C.c_uint8)()
        size = C.c_uint64()
        h = self._handle()
        rc = lib().lacx_encode(h, lp, rp, C.c_uint64(L.size), C.byref(out), C.byref(size))
        if rc != OK:
            _raise(h, rc)
        return _take(out, size)

    def analyze(self, left, right=None):
        """Device analysis only: (block_plans, channel_plans[nblocks*16]) as ctypes arrays."""
        L, lp = _i32(left)
        rp = None
        if right is not None:
            R, rp = _i32(right)
        nb = (L.size + MAX_BLOCK - 1) // MAX_BLOCK
        bplans = (BlockPlan * nb)()
        plans = (ChannelPlan * (nb * SLOTS))()
        h = self._handle()
        rc = lib().lacx_analyze(h, lp, rp, C.c_uint64(L.size), bplans, plans)
        if rc != OK:
            _raise(h, rc)
        return bplans, plans

    def emit_from_plans(self, left, right, bplans, plans) -> bytes:
        """Host-only emit + container from plan records (no device involved)."""
        L, lp = _i32(left)
        rp = None
        if right is not None:
            R, rp = _i32(right)
        out = C.POINTER(C.c_uint8)()
        size = C.c_uint64()
        h = self._handle()
        rc = lib().lacx_emit_from_plans(h, lp, rp, C.c_uint64(L.size), bplans, plans, C.byref(out), C.byref(size))
        if rc != OK:
            _raise(h, rc)
        return _take(out, size)

    def encode_shard(self, left, right=None):
        """Block-range shard: (payload bytes, table uint32[nblocks,2])."""
        L, lp = _i32(left)
        rp = None
        if right is not None:
            R, rp = _i32(right)
        pay = C.POINTER(C.c_uint8)()
        psize = C.c_uint64()
        tab = C.POINTER(C.c_uint32)()
        nb = C.c_uint32()
        h = self._handle()
        rc = lib().lacx_encode_shard(h, lp, rp, C.c_uint64(L.size), C.byref(pay), C.byref(psize), C.byref(tab),
                                     C.byref(nb))
        if rc != OK:
            _raise(h, rc)
        table = np.ctypeslib.as_array(tab, shape=(nb.value, 2)).copy()
        lib().lacx_free(tab)
        return _take(pay, psize), table

    # device-resident entry points (raw pointers; torch tensors welcome) -----------------------
    def encode_device(self, d_left_ptr: int, d_right_ptr: int | None, h_left, h_right, frames: int,
                      stream: int = 0) -> bytes:
        hl, hlp = _i32(h_left)
        hrp = None
        if h_right is not None:
            hr, hrp = _i32(h_right)
        out = C.POINTER(C.c_uint8)()
        size = C.c_uint64()
        h = self._handle()
        rc = lib().lacx_encode_device(h, C.c_void_p(d_left_ptr), C.c_void_p(d_right_ptr or 0), hlp, hrp,
                                      C.c_uint64(frames), C.c_void_p(stream), C.byref(out), C.byref(size))
        if rc != OK:
            _raise(h, rc)
        return _take(out, size)

    def analyze_device(self, d_left_ptr: int, d_right_ptr: int | None, frames: int, stream: int = 0):
        """Runs the kernels on device-resident PCM and returns nothing (plans stay in the encoder)."""
        h = self._handle()
        rc = lib().lacx_analyze_device(h, C.c_void_p(d_left_ptr), C.c_void_p(d_right_ptr or 0), C.c_uint64(frames),
                                       C.c_void_p(stream), None, None)
        if rc != OK:
            _raise(h, rc)

    def encode_shard_device_view(self, d_left_ptr: int, d_right_ptr: int | None, h_left, h_right, frames: int,
                                 stream: int = 0):
        """Zero-copy shard encode: (PayloadView, table) backed by encoder-owned pinned memory."""
        hl, hlp = _i32(h_left)
        hrp = None
        if h_right is not None:
            hr, hrp = _i32(h_right)
        pay = C.POINTER(C.c_uint8)()
        psize = C.c_uint64()
        tab = C.POINTER(C.c_uint32)()
        nb = C.c_uint32()
        h = self._handle()
        rc = lib().lacx_encode_shard_device_view(h, C.c_void_p(d_left_ptr), C.c_void_p(d_right_ptr or 0), hlp, hrp,
                                                 C.c_uint64(frames), C.c_void_p(stream), C.byref(pay),
                                                 C.byref(psize), C.byref(tab), C.byref(nb))
        if rc != OK:
            _raise(h, rc)
        table = np.ctypeslib.as_array(tab, shape=(nb.value, 2))
        return PayloadView(pay, psize.value), table

    def encode_shard_pcm_device_begin(self, data_ptr: int, layout: int, channels: int, frames: int, stream: int = 0,
                                      data1_ptr: int | None = None):
        """Enqueues a shard encode of device-resident PCM and returns at once (see encode_shard_end)."""
        pcm = Pcm(data_ptr, data1_ptr, layout, channels)
        h = self._handle()
        rc = lib().lacx_encode_shard_pcm_device_begin(h, C.byref(pcm), C.c_uint64(frames), C.c_void_p(stream))
        if rc != OK:
            _raise(h, rc)

    def encode_shard_end(self):
        """Waits for the encode started by encode_shard_pcm_device_begin: (PayloadView, table)."""
        pay = C.POINTER(C.c_uint8)()
        psize = C.c_uint64()
        tab = C.POINTER(C.c_uint32)()
        nb = C.c_uint32()
        h = self._handle()
        rc = lib().lacx_encode_shard_end(h, C.byref(pay), C.byref(psize), C.byref(tab), C.byref(nb))
        if rc != OK:
            _raise(h, rc)
        table = np.ctypeslib.as_array(tab, shape=(nb.value, 2))
        return PayloadView(pay, psize.value), table

    def encode_wav(self, wav: bytes) -> bytes:
        """Complete .lac of a PCM WAV file image: the raw data chunk goes to the device as it is (ref
        src/io/wav_io.cpp:167-277 + src/main.cpp:640-675 chained)."""
        out = C.POINTER(C.c_uint8)()
        size = C.c_uint64()
        h = self._handle()
        view = np.frombuffer(wav, dtype=np.uint8)  # no copy: the library only reads the image
        rc = lib().lacx_encode_wav(h, view.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_uint64(view.size), C.byref(out),
                                   C.byref(size))
        if rc != OK:
            _raise(h, rc)
        try:
            return C.string_at(out, size.value)
        finally:
            lib().lacx_free(out)

    def encode_wav_view(self, wav) -> "PayloadView":
        """Zero-copy form of encode_wav: a view of the complete .lac in the encoder's pinned result buffer (valid until
        the encoder's next call); `wav` is any buffer (bytes, numpy uint8 array, mmap)."""
        out = C.POINTER(C.c_uint8)()
        size = C.c_uint64()
        h = self._handle()
        view = np.frombuffer(wav, dtype=np.uint8)
        rc = lib().lacx_encode_wav_view(h, view.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_uint64(view.size), C.byref(out),
                                        C.byref(size))
        if rc != OK:
            _raise(h, rc)
        return PayloadView(out, size.value)

    def encode_shard_pcm_device_view(self, data_ptr: int, layout: int, channels: int, frames: int, stream: int = 0):
        """Zero-copy shard encode of device-resident PCM in its source layout (interleaved int16 / int24)."""
        pcm = Pcm(data_ptr, None, layout, channels)
        pay = C.POINTER(C.c_uint8)()
        psize = C.c_uint64()
        tab = C.POINTER(C.c_uint32)()
        nb = C.c_uint32()
        h = self._handle()
        rc = lib().lacx_encode_shard_pcm_device_view(h, C.byref(pcm), C.c_uint64(frames), C.c_void_p(stream),
                                                     C.byref(pay), C.byref(psize), C.byref(tab), C.byref(nb))
        if rc != OK:
            _raise(h, rc)
        table = np.ctypeslib.as_array(tab, shape=(nb.value, 2))
        return PayloadView(pay, psize.value), table

    def encode_shard_device(self, d_left_ptr: int, d_right_ptr: int | None, h_left, h_right, frames: int,
                            stream: int = 0, copy: bool = True):
        """Shard encode of device-resident PCM. With copy=False the payload comes back as a zero-copy
        `Payload` view of the library's buffer (freed when the object dies)."""
        hl, hlp = _i32(h_left)
        hrp = None
        if h_right is not None:
            hr, hrp = _i32(h_right)
        pay = C.POINTER(C.c_uint8)()
        psize = C.c_uint64()
        tab = C.POINTER(C.c_uint32)()
        nb = C.c_uint32()
        h = self._handle()
        rc = lib().lacx_encode_shard_device(h, C.c_void_p(d_left_ptr), C.c_void_p(d_right_ptr or 0), hlp, hrp,
                                            C.c_uint64(frames), C.c_void_p(stream), C.byref(pay), C.byref(psize),
                                            C.byref(tab), C.byref(nb))
        if rc != OK:
            _raise(h, rc)
        table = np.ctypeslib.as_array(tab, shape=(nb.value, 2)).copy()
        lib().lacx_free(tab)
        if not copy:
            return Payload(pay, psize.value), table
        return _take(pay, psize), table


class Pcm(C.Structure):
    _fields_ = [("data0", C.c_void_p), ("data1", C.c_void_p), ("layout", C.c_uint32), ("channels", C.c_uint32)]


class FanoutShard(C.Structure):
    _fields_ = [("pcm", Pcm), ("frames", C.c_uint64)]


class FanoutOut(C.Structure):
    _fields_ = [("payload", C.POINTER(C.c_uint8)), ("payload_size", C.c_uint64), ("table", C.POINTER(C.c_uint32)),
                ("nblocks", C.c_uint32), ("device", C.c_int32), ("byte_offset", C.c_uint64)]


EXCHANGE_HOST, EXCHANGE_RCCL = 1, 2


class FanoutStats(C.Structure):
    _fields_ = [("lanes_used", C.c_uint32), ("exchange", C.c_uint32), ("exchange_ms", C.c_double), ("concat_ms", C.c_double),
                ("device", C.c_int32 * MAX_FANOUT), ("blocks", C.c_uint32 * MAX_FANOUT), ("lane_frames", C.c_uint64 * MAX_FANOUT),
                ("payload_bytes", C.c_uint64 * MAX_FANOUT), ("encode_ms", C.c_double * MAX_FANOUT)]


def fanout_range(nblocks: int, nlanes: int, lane: int):
    """(first block, block count) of lane `lane` of `nlanes` over a stream of `nblocks` blocks."""
    a, b = C.c_uint32(), C.c_uint32()
    lib().lacx_fanout_range(C.c_uint32(nblocks), C.c_uint32(nlanes), C.c_uint32(lane), C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


PCM_PLANAR_I32, PCM_INTERLEAVED_I16, PCM_INTERLEAVED_I24 = 0, 1, 2


class PayloadView:
    """Borrowed view of encoder-owned pinned memory (valid until the encoder's next call)."""

    def __init__(self, ptr, size):
        self._ptr = ptr
        self.size = size

    def __len__(self):
        return self.size

    def tobytes(self) -> bytes:
        return C.string_at(self._ptr, self.size)


class Payload:
    """Zero-copy view of a malloc'd library buffer."""

    def __init__(self, ptr, size):
        self._ptr = ptr
        self.size = size

    def __len__(self):
        return self.size

    def array(self) -> np.ndarray:
        return np.ctypeslib.as_array(self._ptr, shape=(self.size,)) if self.size else np.zeros(0, np.uint8)

    def tobytes(self) -> bytes:
        return C.string_at(self._ptr, self.size)

    def __del__(self):
        try:
            if self._ptr is not None:
                lib().lacx_free(self._ptr)
                self._ptr = None
        except Exception:
            pass


class WavInfo(C.Structure):
    _fields_ = [("channels", C.c_uint16), ("bit_depth", C.c_uint16), ("sample_rate", C.c_uint32),
                ("frames", C.c_uint64), ("data_offset", C.c_uint64), ("data_bytes", C.c_uint64)]


def wav_parse(wav: bytes):
    """RIFF walk with the reference's accept/reject rules (ref src/io/wav_io.cpp:167-277); None when rejected.
    Host-only: needs no device."""
    info = WavInfo()
    buf = (C.c_uint8 * max(1, len(wav))).from_buffer_copy(wav if wav else b"\0")
    rc = lib().lacx_wav_parse(buf, C.c_uint64(len(wav)), C.byref(info))
    return info if rc == OK else None


class StreamInfo(C.Structure):
    _fields_ = [("sample_rate", C.c_uint32), ("blocks", C.c_uint32), ("frames", C.c_uint64), ("channels", C.c_uint8),
                ("bit_depth", C.c_uint8), ("stereo_mode", C.c_uint8), ("version", C.c_uint8)]


def stream_parse(lac: bytes):
    """Header + block table of a .lac (ref src/codec/lac/decoder.cpp:90-200); None when inconsistent.  Host-only."""
    info = StreamInfo()
    buf = (C.c_uint8 * max(1, len(lac))).from_buffer_copy(lac if lac else b"\0")
    return info if lib().lacx_stream_parse(buf, C.c_uint64(len(lac)), C.byref(info)) == OK else None


def decode(lac: bytes, device: int = -1):
    """LAC::Decoder::decode on the device (ref src/codec/lac/decoder.hpp:10-24): (left, right or None, StreamInfo,
    kernel milliseconds).  Raises RuntimeError("[decode-error] ...") like the reference throws."""
    info = StreamInfo()
    buf = (C.c_uint8 * max(1, len(lac))).from_buffer_copy(lac if lac else b"\0")
    if lib().lacx_stream_parse(buf, C.c_uint64(len(lac)), C.byref(info)) != OK:
        raise RuntimeError(lib().lacx_decode_last_error().decode(errors="replace"))
    left = np.empty(info.frames, dtype=np.int32)
    right = np.empty(info.frames, dtype=np.int32) if info.channels == 2 else None
    ms = C.c_float()
    rc = lib().lacx_decode(C.c_int(device), buf, C.c_uint64(len(lac)), left.ctypes.data_as(C.POINTER(C.c_int32)),
                           right.ctypes.data_as(C.POINTER(C.c_int32)) if right is not None else None,
                           C.c_uint64(info.frames), C.byref(ms))
    if rc != OK:
        raise RuntimeError(lib().lacx_decode_last_error().decode(errors="replace"))
    return left, right, info, float(ms.value)


class Decoder:
    """Mirror of LAC::Decoder (ref src/codec/lac/decoder.hpp:10-24) over a decoder handle (lacx_decoder_create): the device
    buffers live from call to call, and so do the output arrays when `reuse_output` is set (a fresh numpy array of a few
    hundred MB costs more in first-touch page faults than the decode itself)."""

    def __init__(self, device: int = -1, reuse_output: bool = False):
        h = C.c_void_p()
        if lib().lacx_decoder_create(C.c_int(device), C.byref(h)) != OK:
            raise RuntimeError("lacx_decoder_create failed")
        self._h = h
        self._reuse = reuse_output
        self._left = self._right = None

    def close(self):
        if self._h is not None:
            lib().lacx_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode(self, lac: bytes):
        """(left, right or None, StreamInfo, kernel milliseconds); RuntimeError("[decode-error] ...") like the reference."""
        info = StreamInfo()
        buf = np.frombuffer(lac, dtype=np.uint8)
        bp = buf.ctypes.data_as(C.POINTER(C.c_uint8))
        if lib().lacx_stream_parse(bp, C.c_uint64(buf.size), C.byref(info)) != OK:
            raise RuntimeError(lib().lacx_decode_last_error().decode(errors="replace"))
        if self._reuse and self._left is not None and self._left.size == info.frames:
            left, right = self._left, (self._right if info.channels == 2 else None)
        else:
            left = np.empty(info.frames, dtype=np.int32)
            right = np.empty(info.frames, dtype=np.int32) if info.channels == 2 else None
        if right is None and info.channels == 2:
            right = np.empty(info.frames, dtype=np.int32)
        if self._reuse:
            self._left, self._right = left, right
        ms = C.c_float()
        rc = lib().lacx_decoder_decode(self._h, bp, C.c_uint64(buf.size), left.ctypes.data_as(C.POINTER(C.c_int32)),
                                       right.ctypes.data_as(C.POINTER(C.c_int32)) if right is not None else None,
                                       C.c_uint64(info.frames), C.byref(ms))
        if rc != OK:
            raise RuntimeError(lib().lacx_decode_last_error().decode(errors="replace"))
        return left, right, info, float(ms.value)


def assemble(sample_rate: int, bit_depth: int, stereo_mode: int, channels: int, shards) -> bytes:
    """Header + block table + payload concat of (payload, table) shards given in stream order
    (ref src/codec/lac/encoder.cpp:243-250, 445-465)."""
    cfg = Config(sample_rate, bit_depth, stereo_mode, 1, 1, -1, 0, 0)
    n = len(shards)
    pays = (C.c_char_p * n)(*[s[0] for s in shards])
    sizes = (C.c_uint64 * n)(*[len(s[0]) for s in shards])
    tabs_np = [np.ascontiguousarray(s[1], dtype=np.uint32) for s in shards]
    tabs = (C.POINTER(C.c_uint32) * n)(*[t.ctypes.data_as(C.POINTER(C.c_uint32)) for t in tabs_np])
    nbs = (C.c_uint32 * n)(*[t.shape[0] for t in tabs_np])
    out = C.POINTER(C.c_uint8)()
    size = C.c_uint64()
    rc = lib().lacx_assemble(C.byref(cfg), channels, n, C.cast(pays, C.POINTER(C.POINTER(C.c_uint8))), sizes, tabs,
                             nbs, C.byref(out), C.byref(size))
    if rc != OK:
        raise RuntimeError("lacx_assemble failed")
    return _take(out, size)


class BlockEncoder:
    """Mirror of Block::Encoder (ref src/codec/block/encoder.hpp:9-30)."""

    def __init__(self, order: int = 12, debug_lpc: bool = False, debug_zr: bool = False, device: int = -1):
        self._enc = Encoder(order, 0, 48000, 24, device=device)

    def set_zero_run_enabled(self, enabled: bool):
        self._enc.set_zero_run_enabled(enabled)

    def set_partitioning_enabled(self, enabled: bool):
        self._enc.set_partitioning_enabled(enabled)

    def set_debug_block_index(self, index: int):
        pass

    def set_debug_partitions(self, enabled: bool):
        pass

    def encode(self, pcm) -> bytes:
        P, pp = _i32(pcm)
        out = C.POINTER(C.c_uint8)()
        size = C.c_uint64()
        h = self._enc._handle()
        rc = lib().lacx_block_encode(h, pp, C.c_uint32(P.size), C.byref(out), C.byref(size))
        if rc != OK:
            _raise(h, rc)
        return _take(out, size)

    def plan(self, pcm) -> ChannelPlan:
        P, pp = _i32(pcm)
        plan = ChannelPlan()
        h = self._enc._handle()
        rc = lib().lacx_block_plan_only(h, pp, C.c_uint32(P.size), C.byref(plan))
        if rc != OK:
            _raise(h, rc)
        return plan

    def debug_lpc(self, pcm):
        P, pp = _i32(pcm)
        ac = np.zeros(13, dtype=np.int64)
        coef = np.zeros((5, 13), dtype=np.int16)
        used = np.zeros(5, dtype=np.uint8)
        h = self._enc._handle()
        rc = lib().lacx_debug_lpc(h, pp, C.c_uint32(P.size), ac.ctypes.data_as(C.POINTER(C.c_int64)),
                                  coef.ctypes.data_as(C.POINTER(C.c_int16)),
                                  used.ctypes.data_as(C.POINTER(C.c_uint8)))
        if rc != OK:
            _raise(h, rc)
        return ac, coef, used


class BatchItem(C.Structure):
    _fields_ = [("pcm", Pcm), ("frames", C.c_uint64), ("sample_rate", C.c_uint32), ("bit_depth", C.c_uint8),
                ("stereo_mode", C.c_uint8), ("reserved", C.c_uint8 * 2)]


class BatchOut(C.Structure):
    _fields_ = [("payload", C.POINTER(C.c_uint8)), ("payload_size", C.c_uint64), ("table", C.POINTER(C.c_uint32)),
                ("nblocks", C.c_uint32), ("reserved", C.c_uint32)]


class BatchEncoder:
    """Many streams as ONE device job (lacx_encode_batch_device): every stream keeps its own rate, depth, channels and
    stereo mode.  formats: [(sample_rate, bit_depth, stereo_mode), ...] in the order the streams are passed later."""

    def __init__(self, formats, device: int = -1, zero_run: bool = True, partitioning: bool = True):
        self.formats = [tuple(f) for f in formats]
        self._enc = Encoder(12, 2, 48000, 16, device=device)
        self._enc.set_zero_run_enabled(zero_run)
        self._enc.set_partitioning_enabled(partitioning)

    def timing(self) -> Timing:
        return self._enc.timing()

    def encode_device(self, streams, stream: int = 0):
        """streams: [(data_ptr, layout, channels, frames[, data1_ptr]), ...] device-resident PCM; returns a list of
        (PayloadView, table uint32[nblocks, 2]) views into the encoder's pinned result buffer."""
        n = len(streams)
        if n != len(self.formats):
            raise ValueError("one stream per format")
        items = (BatchItem * n)()
        for it, st, (sr, bd, sm) in zip(items, streams, self.formats):
            ptr, layout, ch, frames = st[:4]
            it.pcm = Pcm(ptr, st[4] if len(st) > 4 else None, layout, ch)
            it.frames = frames
            it.sample_rate = sr
            it.bit_depth = bd
            it.stereo_mode = sm
        outs = (BatchOut * n)()
        h = self._enc._handle()
        rc = lib().lacx_encode_batch_device(h, items, C.c_uint32(n), C.c_void_p(stream), outs)
        if rc != OK:
            _raise(h, rc)
        return [(PayloadView(o.payload, o.payload_size), TableView(o.table, o.nblocks)) for o in outs]


class TableView:
    """Borrowed view of a block table (frames, bytes per block) in encoder-owned pinned memory; converted to a numpy array
    only when asked (a batch of many streams returns one per stream on every call)."""

    def __init__(self, ptr, nblocks):
        self._ptr = ptr
        self.shape = (int(nblocks), 2)

    def array(self) -> np.ndarray:
        return np.ctypeslib.as_array(self._ptr, shape=self.shape)

    def copy(self) -> np.ndarray:
        return self.array().copy()

    def __array__(self, dtype=None, copy=None):
        a = self.array()
        return a.astype(dtype) if dtype is not None and a.dtype != dtype else a
