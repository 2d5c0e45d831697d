"""One-off size check on the GPU (not in the default suite): BASELINE configs[3], 2 h of stereo 16-bit 48 kHz
(345.6 M frames, 21 094 blocks, 1.38 GB of PCM) encoded in ONE call on one GPU.  The last eighth of the result
(blocks 18457..21093) must be byte-identical to the golden digest of that shard minted from the reference."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_pkg()
lacx, synth = pkg.lacx, pkg.synth
frames, sr, bd, sm = 345_600_000, 48000, 16, 2
ent = [e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "digests.json"))) if e["name"].startswith("cfg4_2h_shard8of8")][0]
t0 = time.time()
d = torch.empty(frames * 2, dtype=torch.int16, device="cuda")
step = 16384 * 1024
for f0 in range(0, frames, step):  # generate and upload piecewise: bounded host memory
    n = min(step, frames - f0)
    l, r = synth.synth_pcm(n, 2, bd, sr, seed=2026, kind="music", stereo="wide", start=f0)
    d[2 * f0:2 * (f0 + n)] = torch.from_numpy(synth.interleave(l, r, bd).view(np.int16)).cuda()
print(f"generated {frames} frames in {time.time() - t0:.0f} s", flush=True)
enc = lacx.Encoder(12, sm, sr, bd, device=0)
for it in range(2):
    t0 = time.time()
    payload, table = enc.encode_shard_pcm_device_view(d.data_ptr(), lacx.PCM_INTERLEAVED_I16, 2, frames, 0)
    dt = time.time() - t0
    print(f"encode {it}: {dt * 1e3:.1f} ms -> {frames * 2 / dt / 1e6:.0f} Msamples/s, payload {payload.size} B, {table.shape[0]} blocks", flush=True)
for nch in (os.environ.get("BIG_SWEEP", "").split(",") if os.environ.get("BIG_SWEEP") else []):
    os.environ["LACX_PIPE_CHUNKS"] = nch
    best = 1e9
    for it in range(3):
        t0 = time.time()
        payload, table = enc.encode_shard_pcm_device_view(d.data_ptr(), lacx.PCM_INTERLEAVED_I16, 2, frames, 0)
        best = min(best, time.time() - t0)
    print(f"chunks {nch}: {best * 1e3:.1f} ms -> {frames * 2 / best / 1e6:.0f} Msamples/s", flush=True)
os.environ.pop("LACX_PIPE_CHUNKS", None)
b0 = 18457
assert table.shape[0] == 21094
off = int(table[:b0, 1].astype(np.int64).sum())
tail = payload.tobytes()[off:]
lac = lacx.assemble(sr, bd, sm, 2, [(tail, table[b0:].copy())])
ok = len(lac) == ent["lac_bytes"] and hashlib.sha256(lac).hexdigest() == ent["lac_sha256"]
print("last eighth identical to the reference's shard:", ok)
if os.environ.get("BIG_DECODE", "1") != "0":
    # the whole 2 h stream through the device decoder: 21 094 lanes, one per block, all at once
    whole = lacx.assemble(sr, bd, sm, 2, [(payload.tobytes(), table.copy())])
    dl, dr, info, ms = lacx.decode(whole)
    pcm = d.cpu().numpy().reshape(-1, 2)
    same = bool(np.array_equal(dl, pcm[:, 0]) and np.array_equal(dr, pcm[:, 1]))
    print(f"decode: kernels {ms:.1f} ms for {frames * 2 / 1e6:.0f} Msamples = {frames * 2 / ms / 1e3:.0f} Msamples/s, "
          f"{info.blocks} blocks, PCM identical: {same}", flush=True)
    ok = ok and same
sys.exit(0 if ok else 1)
