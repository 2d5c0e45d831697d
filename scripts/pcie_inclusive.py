"""Diagnostic: whole-call rates with the PCM starting in host memory (never the bench's `value`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as ge
import wavutil as W
pkg = ge.load_pkg(); lacx, synth = pkg.lacx, pkg.synth
frames = 28_800_000
L, R = synth.synth_pcm(frames, 2, 16, 48000, seed=2026, kind="music")
enc = lacx.Encoder(12, 2, 48000, 16)
wav = W.make_wav(L, R, 48000, 16)
for name, fn in (("lacx_encode (planar int32 host PCM, 230 MB H2D)", lambda: enc.encode(L, R)),
                 ("lacx_encode_wav (WAV image in host memory, 115 MB H2D)", lambda: enc.encode_wav(wav)),
                 ("lacx_encode again", lambda: enc.encode(L, R))):
    fn()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); out = fn(); best = min(best, time.perf_counter() - t0)
    t = enc.timing()
    print(f"{name}: {best * 1e3:.1f} ms from Python (incl. the bytes copy of the result) -> {frames * 2 / best / 1e6:.0f} Msamples/s; "
          f"inside the library {t.total_ms:.1f} ms of which H2D {t.h2d_ms:.1f} ms ({len(out)} B)")
for devs in ([0, 0], [0, 0, 0, 0]):
    fenc = lacx.Encoder(12, 2, 48000, 16, devices=devs)
    fenc.encode(L, R)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); out2 = fenc.encode(L, R); best = min(best, time.perf_counter() - t0)
    st = fenc.fanout_stats()
    print(f"lacx_encode over devices {devs}: {best * 1e3:.1f} ms from Python; lanes {st.lanes_used}, lane encode ms {[round(st.encode_ms[g], 2) for g in range(st.lanes_used)]}, "
          f"exchange {st.exchange_ms:.2f} ms, concat {st.concat_ms:.2f} ms, uploader {fenc.timing().h2d_ms:.2f} ms; same bytes {out2 == out if len(out2) == len(out) else False}")
