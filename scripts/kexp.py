"""Diagnostic A/B harness (not part of the default suite): times the device-resident encode of one workload with
differently built libraries, one child process per library (KEXP_LIB -> lacx.use_library), and checks the bytes against the golden
digest where one exists.  usage: kexp.py <lib.so>[,<lib.so>...] [seconds=600] [kind=music] [bit_depth=16] [rate=48000]
A library whose name contains "stamps" also prints the in-kernel phase stamps (scripts/stamps.py)."""
import hashlib, json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(secs, kind, bd, sr):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    lacx, synth = pkg.lacx, pkg.synth
    lacx.use_library(os.environ["KEXP_LIB"])
    frames = secs * sr
    seed, stereo = (2026, "wide") if kind == "music" else ((7, "wide") if kind == "mixed" else (3, "independent"))
    L, R = synth.synth_pcm(frames, 2, bd, sr, seed=seed, kind=kind, stereo=stereo)
    inter = synth.interleave(L, R, bd)
    d = torch.from_numpy(inter.view(np.int16) if bd == 16 else inter).cuda()
    enc = lacx.Encoder(12, 2, sr, bd, device=0)
    layout = lacx.PCM_INTERLEAVED_I16 if bd == 16 else lacx.PCM_INTERLEAVED_I24
    ablation = bool(os.environ.get("LACX_DEBUG_SKIP"))  # timing ablations produce wrong plans: errors are expected

    def once():
        enc.encode_shard_pcm_device_begin(d.data_ptr(), layout, 2, frames, 0)
        try:
            return enc.encode_shard_end()
        except RuntimeError:
            if not ablation:
                raise
            return None

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    full, step, probe, ingest = [], [], [], []
    res = None
    for _ in range(10):
        t0 = time.perf_counter()
        res = once()
        step.append((time.perf_counter() - t0) * 1e3)
        tm = enc.timing()
        full.append(tm.full_ms)
        probe.append(tm.probe_ms)
        ingest.append(tm.ingest_ms)
    sha = None
    if res is not None:
        pay, tab = res
        lac = lacx.assemble(sr, bd, 2, 2, [(pay.tobytes(), np.array(tab, dtype=np.uint32))])
        sha = hashlib.sha256(lac).hexdigest()
    with open(os.path.join(ROOT, "tests", "golden", "digests.json")) as f:
        dg = {(e["gen"]["frames"], e["gen"]["bit_depth"], e["gen"]["sample_rate"], e["gen"]["kind"], e["gen"]["seed"]): e for e in json.load(f)
              if e["gen"]["channels"] == 2 and e["stereo_mode"] == 2 and e["gen"].get("start", 0) == 0}
    ent = dg.get((frames, bd, sr, kind, seed))
    ok = None if ent is None else (ent["lac_sha256"] == sha)
    print(f"RESULT full_ms min {min(full):.4f} med {sorted(full)[5]:.4f}  step_ms min {min(step):.3f} med {sorted(step)[5]:.3f}  probe_ms med {sorted(probe)[5]:.4f} ingest_ms med {sorted(ingest)[5]:.4f}  digest {ok}", flush=True)
    if "stamps" in os.environ.get("KEXP_LIB", ""):
        import ctypes as C
        buf = (C.c_ulonglong * 40)()
        lacx.lib().lacx_debug_stamps(buf)
        names = ["stage", "score+select", "pass1: all bounds", "resid+store+scan1", "Bsel wait", "pass1 barrier", "scan2+planes", "part: quick", "phase_a", "part: queued", "B3 wait", "quick+enqueue", "phase_b", "reduce", "B5 wait", "-", "part: r+scan", "grp+scan", "seg_static", "part pass", "B wait", "choose+final", "(realtime)", "emit: plan+nx scan", "emit: phase A", "emit: walk 1", "emit: bit scan", "emit: zero tile", "emit: walk 2", "emit: barrier", "emit: copy+publish", "-"]
        idx = [i for i in range(32) if i != 22]
        tot = sum(buf[i] for i in idx)
        waves = max(1, buf[32])
        print(f"  waves {waves} cycles/wave {tot / waves:.0f} lifetime {buf[22] / waves / 100:.1f} us")
        for i in idx:
            if buf[i]:
                print(f"  {names[i]:22s} {buf[i] / waves:10.0f} cyc/wave {100.0 * buf[i] / max(1, tot):5.1f}%")


if __name__ == "__main__":
    if os.environ.get("KEXP_CHILD"):
        child(int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    libs = sys.argv[1].split(",")
    rest = sys.argv[2:] + ["600", "music", "16", "48000"][len(sys.argv) - 2:]
    for lib in libs:
        env = dict(os.environ, KEXP_CHILD="1", KEXP_LIB=os.path.join(ROOT, lib))
        print(f"== {lib} {' '.join(rest)}", flush=True)
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + rest, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        print("\n".join(l for l in p.stdout.splitlines() if l.startswith("RESULT") or l.startswith("  ")) or p.stdout[-2000:], flush=True)
