"""Measurement (not part of the default suite): BASELINE configs[4] as one batched job on one GPU -- the 16 combinations
{mono, stereo} x {16, 24 bit} x {44.1, 48, 96, 192 kHz}, 60 s each, PCM resident in HBM in its WAV layout.  One encoder
per stream; "serial" encodes them one after the other, "batched" enqueues all sixteen (lacx_encode_shard_pcm_device_begin)
before it collects any (lacx_encode_shard_end), so that the small jobs' kernels share the chip.  Every result is
compared with the oracle's bytes once."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import __graft_entry__ as ge
import oracleshim

pkg = ge.load_pkg()
lacx, synth = pkg.lacx, pkg.synth
secs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
check = len(sys.argv) <= 2 or sys.argv[2] != "nocheck"
jobs = []
for ch in (1, 2):
    for bd in (16, 24):
        for sr in (44100, 48000, 96000, 192000):
            frames = secs * sr
            left, right = synth.synth_pcm(frames, ch, bd, sr, seed=500 + len(jobs), kind="mixed" if len(jobs) & 1 else "music")
            inter = synth.interleave(left, right, bd)
            arr = inter.view(np.int16) if bd == 16 else inter
            d = torch.from_numpy(np.ascontiguousarray(arr)).cuda()
            sm = 2 if ch == 2 else 0
            layout = lacx.PCM_INTERLEAVED_I16 if bd == 16 else lacx.PCM_INTERLEAVED_I24
            jobs.append(dict(ch=ch, bd=bd, sr=sr, sm=sm, frames=frames, d=d, layout=layout, left=left, right=right,
                             enc=lacx.Encoder(12, sm, sr, bd, device=0)))
total_samples = sum(j["frames"] * j["ch"] for j in jobs)


def serial():
    out = []
    for j in jobs:
        p, t = j["enc"].encode_shard_pcm_device_view(j["d"].data_ptr(), j["layout"], j["ch"], j["frames"])
        out.append((p, t))
    return out


def batched():
    for j in jobs:
        j["enc"].encode_shard_pcm_device_begin(j["d"].data_ptr(), j["layout"], j["ch"], j["frames"])
    return [j["enc"].encode_shard_end() for j in jobs]


for name, fn in (("serial", serial), ("batched", batched)):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        res = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{name:8s} {best * 1e3:8.2f} ms for 16 streams of {secs} s = {total_samples / best / 1e6:9.1f} Msamples/s", flush=True)
    if check:
        bad = 0
        for j, (p, t) in zip(jobs, res):
            got = lacx.assemble(j["sr"], j["bd"], j["sm"], j["ch"], [(p.tobytes(), t.copy())])
            want = oracleshim.encode(j["left"], j["right"], j["sr"], j["bd"], j["sm"], threads=8)
            bad += got != want
        print(f"         {16 - bad}/16 byte-identical to the oracle", flush=True)
