"""GPU tests of the multi-device fan-out behind the C ABI (ref src/codec/lac/encoder.cpp:385-465):
an encoder over a device list cuts the stream into contiguous block ranges, one lane (host thread + encoder) per list
entry, exchanges the shard sizes and concatenates on the host.  The bytes must not depend on the list.  A one-GPU box
rehearses with repeated ordinals ([0, 0], [0, 0, 0, 0]); with several GPUs visible the list of all of them runs too."""
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


def device_lists(gpu):
    lists = [[0], [0, 0], [0, 0, 0, 0], [0, 0, 0]]
    n = gpu.lacx.device_count()
    if n > 1:
        lists.append(list(range(n)))
        lists.append(list(range(n)) * 2)
    return lists


def test_ragged_stream_over_device_lists(gpu, oracle):
    """A ragged stereo stream (per-block auto stereo, mixed material, small final block) and a mono 24-bit one over every
    device list: bytes equal to the oracle's and to the single-device call's; the statistics name the lanes."""
    for frames, ch, bd, sr, sm, kind, seed in ((16384 * 13 + 1234, 2, 16, 48000, 2, "mixed", 21), (16384 * 9 + 5, 1, 24, 96000, 0, "music", 22),
                                                (16384 * 3 + 300, 2, 24, 44100, 2, "noise", 23)):
        left, right = gpu.synth.synth_pcm(frames, ch, bd, sr, seed=seed, kind=kind)
        want = oracle.encode(left, right, sr, bd, sm, threads=8)
        single = gpu.lacx.Encoder(12, sm, sr, bd, device=0).encode(left, right)
        assert single == want
        nb = -(-frames // 16384)
        for devs in device_lists(gpu):
            enc = gpu.lacx.Encoder(12, sm, sr, bd, devices=devs, min_blocks_per_device=1)
            for _ in range(2):
                assert enc.encode(left, right) == want, (devs, frames)
            assert enc.lanes() == len(devs)
            if len(devs) > 1:
                st = enc.fanout_stats()
                used = min(len(devs), nb)
                assert st.lanes_used == used
                assert [st.device[g] for g in range(used)] == devs[:used]
                assert sum(st.blocks[g] for g in range(used)) == nb
                assert [st.blocks[g] for g in range(used)] == [gpu.lacx.fanout_range(nb, used, g)[1] for g in range(used)]
                assert sum(st.payload_bytes[g] for g in range(used)) == len(want) - 14 - 8 * nb
                distinct = len(set(devs)) == len(devs)
                assert st.exchange == (gpu.lacx.EXCHANGE_RCCL if distinct else gpu.lacx.EXCHANGE_HOST), enc.fanout_exchange_note()


def test_eighth_of_the_two_hour_stream_over_device_lists(gpu):
    """BASELINE configs[3]: one eighth of the 2 h stream (2 637 blocks) against the digest minted from the reference, over
    every device list, with the default lane size (64 blocks) and through the WAV entry points as well."""
    import wavutil as W

    with open(os.path.join(GOLDEN, "digests.json")) as f:
        ent = {e["name"]: e for e in json.load(f)}["cfg4_2h_shard3of8_st16_48k"]
    g = ent["gen"]
    left, right = gpu.synth.synth_pcm(g["frames"], g["channels"], g["bit_depth"], g["sample_rate"], seed=g["seed"], kind=g["kind"],
                                      stereo=g.get("stereo", "wide"), start=g.get("start", 0))
    wav = W.make_wav(left, right, g["sample_rate"], g["bit_depth"])
    for devs in device_lists(gpu):
        enc = gpu.lacx.Encoder(12, ent["stereo_mode"], g["sample_rate"], g["bit_depth"], devices=devs)
        got = enc.encode(left, right)
        assert len(got) == ent["lac_bytes"] and hashlib.sha256(got).hexdigest() == ent["lac_sha256"], devs
        t = enc.timing()
        assert t.full_slots == 2 * -(-g["frames"] // 16384)
        view = enc.encode_wav_view(wav)
        assert view.tobytes() == got, devs
        assert enc.encode_wav(wav) == got, devs


def test_default_lane_size_keeps_short_streams_on_fewer_devices(gpu, oracle):
    """A lane is only used when every lane gets at least min_blocks_per_device blocks (default 64): 100 blocks over four
    lanes run on one, 200 blocks on three."""
    for nblocks, want_lanes in ((100, 1), (200, 3)):
        left, right = gpu.synth.synth_pcm(16384 * nblocks, 2, 16, 48000, seed=5, kind="music")
        want = oracle.encode(left, right, 48000, 16, 2, threads=8)
        enc = gpu.lacx.Encoder(12, 2, 48000, 16, devices=[0, 0, 0, 0])
        assert enc.encode(left, right) == want
        assert enc.fanout_stats().lanes_used == want_lanes


def test_host_emit_and_switches_through_the_fanout(gpu, oracle):
    """The north_star layout (bit emit on host threads) and the zero-run / partitioning switches reach every lane."""
    left, right = gpu.synth.synth_pcm(16384 * 6 + 77, 2, 16, 48000, seed=31, kind="mixed")
    for host_emit, zr, part in ((True, True, True), (False, False, True), (True, True, False)):
        want = oracle.encode(left, right, 48000, 16, 2, threads=8, zero_run=zr, partitioning=part)
        enc = gpu.lacx.Encoder(12, 2, 48000, 16, devices=[0, 0, 0], min_blocks_per_device=1)
        enc.set_host_emit(host_emit)
        enc.set_zero_run_enabled(zr)
        enc.set_partitioning_enabled(part)
        assert enc.encode(left, right) == want, (host_emit, zr, part)


def test_sample_range_errors_name_the_stream_index(gpu):
    """The reference reports the first bad LEFT sample anywhere in the stream, then the first bad right one (ref
    lac/encoder.cpp:238-241): with a right error in an early shard and a left error in a later one, left wins."""
    frames = 16384 * 8 + 100
    left, right = gpu.synth.synth_pcm(frames, 2, 16, 48000, seed=8, kind="music")
    enc = gpu.lacx.Encoder(12, 2, 48000, 16, devices=[0, 0, 0, 0], min_blocks_per_device=1)
    single = gpu.lacx.Encoder(12, 2, 48000, 16, device=0)
    bad_r = right.copy()
    bad_r[16384 + 7] = -40000            # lane 0
    bad_l = left.copy()
    bad_l[16384 * 6 + 11] = 70000        # lane 3
    bad_l[16384 * 7 + 1] = 70000
    for e in (enc, single):
        with pytest.raises(ValueError, match=rf"left sample at index {16384 * 6 + 11} is outside"):
            e.encode(bad_l, bad_r)
        with pytest.raises(ValueError, match=rf"right sample at index {16384 + 7} is outside"):
            e.encode(left, bad_r)
    assert enc.encode(left, right) == single.encode(left, right)   # and the encoder is usable afterwards


def test_resident_shards(gpu, oracle):
    """lacx_encode_fanout_resident: the shards already in device memory (the bench's timed region); views + byte offsets
    assemble to the stream's bytes."""
    import torch

    frames = 16384 * 11 + 999
    left, right = gpu.synth.synth_pcm(frames, 2, 16, 48000, seed=77, kind="mixed")
    want = oracle.encode(left, right, 48000, 16, 2, threads=8)
    nb = -(-frames // 16384)
    for devs in ([0, 0], [0, 0, 0, 0]) + (tuple([list(range(gpu.lacx.device_count()))]) if gpu.lacx.device_count() > 1 else ()):
        enc = gpu.lacx.Encoder(12, 2, 48000, 16, devices=devs)
        shards, keep = [], []
        for g, dev in enumerate(devs):
            b0, cnt = gpu.lacx.fanout_range(nb, len(devs), g)
            f0, f1 = b0 * 16384, min(frames, (b0 + cnt) * 16384)
            inter = gpu.synth.interleave(left[f0:f1], right[f0:f1], 16)
            d = torch.from_numpy(inter.view(np.int16)).to(f"cuda:{dev}")
            keep.append(d)
            shards.append((d.data_ptr(), gpu.lacx.PCM_INTERLEAVED_I16, 2, f1 - f0))
        torch.cuda.synchronize()
        for _ in range(2):
            res = enc.encode_fanout_resident(shards)
            off = 0
            for (pay, tab, dev, byte_off), want_dev in zip(res, devs):
                assert dev == want_dev and byte_off == off
                off += len(pay)
            got = gpu.lacx.assemble(48000, 16, 2, 2, [(p.tobytes(), np.array(t, dtype=np.uint32)) for p, t, _, _ in res])
            assert got == want, devs


def test_rccl_exchange_runs_on_one_device(gpu, oracle, monkeypatch):
    """LACX_FANOUT_EXCHANGE=rccl with a list of one device keeps the whole fan-out machinery (one lane): librccl.so is
    loaded, ncclCommInitAll creates the communicator, the lane's sizes go through ncclAllGather on its stream."""
    monkeypatch.setenv("LACX_FANOUT_EXCHANGE", "rccl")
    left, right = gpu.synth.synth_pcm(16384 * 5 + 3, 2, 16, 48000, seed=9, kind="music")
    want = oracle.encode(left, right, 48000, 16, 2, threads=8)
    enc = gpu.lacx.Encoder(12, 2, 48000, 16, devices=[0])
    for _ in range(3):
        assert enc.encode(left, right) == want
    st = enc.fanout_stats()
    assert st.lanes_used == 1 and st.exchange == gpu.lacx.EXCHANGE_RCCL, enc.fanout_exchange_note()
    monkeypatch.setenv("LACX_FANOUT_EXCHANGE", "host")
    enc2 = gpu.lacx.Encoder(12, 2, 48000, 16, devices=[0, 0], min_blocks_per_device=1)
    assert enc2.encode(left, right) == want
    assert enc2.fanout_stats().exchange == gpu.lacx.EXCHANGE_HOST


def test_all_devices_is_the_default_of_the_cpp_mirror(gpu):
    """LAC::Encoder (the C++ mirror) spreads a stream over every visible device by itself and takes an explicit list
    through set_devices: tests/native/mirror_api_test.cpp checks list-independence of the bytes."""
    import subprocess

    from test_host_side import _build_mirror_test

    assert subprocess.call([_build_mirror_test(), "fanout"]) == 0
