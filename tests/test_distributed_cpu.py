"""world_size-2 gloo test of the multi-GPU data path: frames are sharded over ranks with no
data-path collective; the only communication is the bench's barrier and the max-over-ranks time."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from popsift_amd.dispatch import shard_range
    b, e = shard_range(n_frames, world, rank)
    mine = torch.zeros(n_frames, dtype=torch.int64)
    mine[b:e] = 1                          # "processed" frames of this rank (no exchange needed)
    t_local = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t_local, op=dist.ReduceOp.MAX)      # bench.py: time = max over ranks
    cover = mine.clone()
    dist.all_reduce(cover, op=dist.ReduceOp.SUM)        # test-only: who processed what
    q.put((rank, b, e, float(t_local.item()), cover.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [64, 7])
def test_two_rank_sharding(n_frames):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n_frames     # contiguous partition
    assert all(abs(r[3] - 0.020) < 1e-12 for r in res)                             # max over ranks
    assert res[0][4] == [1] * n_frames                                             # every frame exactly once
