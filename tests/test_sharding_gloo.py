"""world_size-2 gloo rehearsal of the multi-GPU path: contiguous block-range split, the all_gather of shard sizes
(the path's only exchange step) and assembly in rank order.  On CPU (no GPU in the build container) the shard encoder
is the oracle (test infrastructure) and the test covers the host logic; the `gpu` variant runs the same flow with
every rank calling the product's lacx_encode_shard on the HIP device, and the bench's own `--gpus 2` launcher."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, frames, tmpdir, use_gpu=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as ge

    pkg = ge.load_pkg()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blocks = (frames + 16383) // 16384
    b0, b1 = rank * blocks // world, (rank + 1) * blocks // world
    f0, f1 = b0 * 16384, min(b1 * 16384, frames)
    left, right = pkg.synth.synth_pcm(f1 - f0, 2, 16, 48000, seed=31, kind="mixed", start=f0)
    if use_gpu:  # the product path: HIP kernels behind the C ABI's shard entry point
        payload, table = pkg.lacx.Encoder(12, 2, 48000, 16, device=0).encode_shard(left, right)
        nb = table.shape[0]
    else:
        import oracleshim

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


def _split_case(tmp_path, pkg, oracle, use_gpu):
    frames = 16384 * 5 + 700
    left, right = pkg.synth.synth_pcm(frames, 2, 16, 48000, seed=31, kind="mixed")
    whole = oracle.encode(left, right, 48000, 16, 2, threads=4)
    nb = int.from_bytes(whole[10:14], "big")
    total_payload = len(whole) - 14 - 8 * nb
    with open(tmp_path / "payload.bin", "wb") as f:
        f.truncate(total_payload)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, frames, str(tmp_path), use_gpu), nprocs=2, join=True)
    payload = open(tmp_path / "payload.bin", "rb").read()
    tables = [np.load(tmp_path / f"table{r}.npy") for r in range(2)]
    sizes = [int(t[:, 1].sum()) for t in tables]
    shards = [(payload[:sizes[0]], tables[0]), (payload[sizes[0]:], tables[1])]
    assert pkg.lacx.assemble(48000, 16, 2, 2, shards) == whole


def test_two_rank_block_range_split(tmp_path, pkg, oracle):
    _split_case(tmp_path, pkg, oracle, use_gpu=False)


@pytest.mark.gpu
def test_two_rank_block_range_split_on_the_hip_path(tmp_path, pkg, oracle):
    """Same flow, every rank encoding its block range with lacx_encode_shard on the GPU (two ranks share device 0)."""
    if pkg.lacx.device_count() < 1:
        pytest.fail("no HIP device visible")
    _split_case(tmp_path, pkg, oracle, use_gpu=True)


@pytest.mark.gpu
def test_bench_launches_its_own_ranks(pkg):
    """`python bench.py --gpus 2` with no WORLD_SIZE starts the ranks itself and relays one JSON line (on a one-GPU box
    the ranks share the GPU: rehearsal mode)."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--seconds", "20", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["value"] > 0
