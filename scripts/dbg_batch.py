import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_pkg(); lacx, synth = pkg.lacx, pkg.synth
secs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
jobs = []
for ch in (1, 2):
    for bd in (16, 24):
        for sr in (44100, 48000, 96000, 192000):
            i = len(jobs)
            l, r = synth.synth_pcm(secs * sr, ch, bd, sr, seed=500 + i, kind="mixed" if i & 1 else "music")
            inter = synth.interleave(l, r, bd)
            d = torch.from_numpy(inter.view(np.int16) if bd == 16 else inter).cuda()
            jobs.append((d, lacx.PCM_INTERLEAVED_I16 if bd == 16 else lacx.PCM_INTERLEAVED_I24, ch, secs * sr, sr, bd, 2 if ch == 2 else 0))
be = lacx.BatchEncoder([(j[4], j[5], j[6]) for j in jobs], device=0)
for rep in range(4):
    try:
        res = be.encode_device([(j[0].data_ptr(), j[1], j[2], j[3]) for j in jobs], torch.cuda.current_stream().cuda_stream)
        t = be.timing()
        print("rep", rep, "ok", sum(len(p) for p, _ in res), "full_ms", round(t.full_ms, 3), "total_ms", round(t.total_ms, 3), "gave_up", t.packer_gave_up, "repacked", t.moved_by_k_pack, flush=True)
    except RuntimeError as e:
        print("rep", rep, "ERR", e, flush=True)
        import ctypes as C
