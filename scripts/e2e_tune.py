"""Diagnostic: host WAV image -> host .lac (lacx_encode_wav_view) for a list of upload pipeline shapes (LACX_PIPE_SPLIT
values; knobs are read when an encoder is created).  usage: e2e_tune.py "1,3,4" "1,2,3,4,6" ...   ("" = the default)"""
import os, subprocess, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2:  # one child process per shape: the process's first encoder is measurably faster than later ones
    for a in sys.argv[1:]:
        subprocess.run([sys.executable, os.path.abspath(__file__), a])
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
import wavutil as W
pkg = ge.load_pkg(); lacx, synth = pkg.lacx, pkg.synth
bd, sr, kind, seed = (16, 48000, "music", 2026) if os.environ.get("E2E_FMT", "16") == "16" else (24, 96000, "mixed", 7)
frames = 600 * sr
cache = f"/tmp/e2e_tune_{bd}_{sr}.npy"
if os.path.exists(cache):
    wav = np.load(cache)
else:
    L, R = synth.synth_pcm(frames, 2, bd, sr, seed=seed, kind=kind)
    wav = np.frombuffer(W.make_wav(L, R, sr, bd), dtype=np.uint8)
    np.save(cache, wav)
ref = None
for split in sys.argv[1:] or [""]:
    if split:
        os.environ["LACX_PIPE_SPLIT"] = split
    else:
        os.environ.pop("LACX_PIPE_SPLIT", None)
    enc = lacx.Encoder(12, 2, sr, bd, device=0)
    enc.encode_wav_view(wav); enc.encode_wav_view(wav)
    ts, h2d = [], []
    for _ in range(8):
        v = enc.encode_wav_view(wav); t = enc.timing(); ts.append(t.total_ms); h2d.append(t.h2d_ms)
    d = hashlib.sha256(v.tobytes()).hexdigest()[:12]
    ref = ref or d
    print(f"split {split or 'default':14s} total_ms min {min(ts):.3f} med {sorted(ts)[4]:.3f}  uploader_ms {min(h2d):.3f}  analysis_ms {t.analysis_ms:.3f} full_ms {t.full_ms:.3f} launches {t.full_launches}  enqueue {t.enqueue_ms:.2f} drain first {t.drain_first_ms:.2f} last {t.drain_last_ms:.2f} copies {t.drain_copies} kernels_done {t.kernels_done_ms:.2f} pollgap {t.poll_gap_max_ms:.2f} same_bytes {d == ref}", flush=True)
    enc.close()
