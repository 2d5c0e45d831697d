"""world_size-2 gloo rehearsal of the multi-GPU path's host logic: contiguous block-range split, the
all_gather of shard sizes (the path's only exchange step) and assembly in rank order.  The shard encoder
here is the oracle (test infrastructure); on the GPU box bench.py uses the HIP path for the same flow."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, frames, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as ge
    import oracleshim

    pkg = ge.load_pkg()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blocks = (frames + 16383) // 16384
    b0, b1 = rank * blocks // world, (rank + 1) * blocks // world
    f0, f1 = b0 * 16384, min(b1 * 16384, frames)
    left, right = pkg.synth.synth_pcm(f1 - f0, 2, 16, 48000, seed=31, kind="mixed", start=f0)
    data = oracleshim.encode(left, right, 48000, 16, 2)
    nb = int.from_bytes(data[10:14], "big")
    table = np.frombuffer(data[14:14 + 8 * nb], dtype=">u4").reshape(nb, 2).astype(np.uint32)
    payload = data[14 + 8 * nb:]
    mine = torch.tensor([len(payload), nb], dtype=torch.int64)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    offset = sum(int(v[0]) for v in allv[:rank])
    assert sum(int(v[1]) for v in allv) == blocks
    np.save(os.path.join(tmpdir, f"table{rank}.npy"), table)
    with open(os.path.join(tmpdir, "payload.bin"), "r+b") as f:  # host-side concat by byte offset
        f.seek(offset)
        f.write(payload)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_block_range_split(tmp_path, pkg, oracle):
    frames = 16384 * 5 + 700
    left, right = pkg.synth.synth_pcm(frames, 2, 16, 48000, seed=31, kind="mixed")
    whole = oracle.encode(left, right, 48000, 16, 2, threads=4)
    nb = int.from_bytes(whole[10:14], "big")
    total_payload = len(whole) - 14 - 8 * nb
    with open(tmp_path / "payload.bin", "wb") as f:
        f.truncate(total_payload)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, frames, str(tmp_path)), nprocs=2, join=True)
    payload = open(tmp_path / "payload.bin", "rb").read()
    tables = [np.load(tmp_path / f"table{r}.npy") for r in range(2)]
    sizes = [int(t[:, 1].sum()) for t in tables]
    shards = [(payload[:sizes[0]], tables[0]), (payload[sizes[0]:], tables[1])]
    assert pkg.lacx.assemble(48000, 16, 2, 2, shards) == whole
