"""Diagnostic: repeated encodes of varying size through every entry point; host RSS and device memory must level off."""
import os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_pkg(); lacx, synth = pkg.lacx, pkg.synth
rng = np.random.default_rng(3)
L, R = synth.synth_pcm(16384 * 40, 2, 16, 48000, seed=5, kind="mixed")
inter = torch.from_numpy(synth.interleave(L, R, 16).view(np.int16)).cuda()
def rss(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024
for rnd in range(6):
    for it in range(60):
        n = int(rng.integers(1, L.size))
        enc = lacx.Encoder(12, int(rng.integers(0, 3)), 48000, 16)
        enc.set_host_emit(bool(it & 1))
        lac = enc.encode(L[:n], R[:n])
        if it % 5 == 0:  # and back through the device decoder (allocates and frees its buffers per call)
            dl, dr, _, _ = lacx.decode(lac)
            assert np.array_equal(dl, L[:n]) and np.array_equal(dr, R[:n])
        if it % 3 == 0:
            enc.set_host_emit(False)
            enc.encode_shard_pcm_device_view(inter.data_ptr(), lacx.PCM_INTERLEAVED_I16, 2, n, 0)
        del enc
    free, total = torch.cuda.mem_get_info()
    print(f"round {rnd}: max RSS {rss()} MiB, device used {(total - free) >> 20} MiB", flush=True)
