"""N>1 path on CPU (gloo, world_size 2): the ray-band sharding of bench.py covers every row exactly once and the
flat gradient all-reduce + average reproduces the single-process gradient of the mean loss."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the sharding / all-reduce helpers of the real bench, not a copy)


def bands(res, world):
    return [bench.row_band(r, world, res) for r in range(world)]


def test_row_bands_partition_the_frame():
    for res, world in [(512, 1), (512, 2), (512, 8), (500, 8), (64, 3)]:
        b = bands(res, world)
        assert b[0][0] == 0 and b[-1][1] == res
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from oracle import lab4d_oracle as O
    torch.manual_seed(0)
    W = torch.randn(8, 63, requires_grad=True)          # same weights on every rank
    g = torch.Generator().manual_seed(1)
    x_all = torch.randn(64, 3, generator=g)
    r0, r1 = bands(64, world)[rank]
    loss = (O.pos_embedding(x_all[r0:r1], 10) @ W.t()).pow(2).mean()   # per-rank normaliser (DDP semantics)
    b = torch.randn(5, requires_grad=True)              # a second parameter: the flat buffer must be split back correctly
    loss = loss + (b * (rank + 1)).sum()
    loss.backward()
    bench.allreduce_grads([W, b], world)
    q.put((rank, torch.cat([W.grad.reshape(-1), b.grad])))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_matches_single_process():
    world, port = 2, 29561
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert torch.allclose(res[0], res[1])
    sys.path.insert(0, ROOT)
    from oracle import lab4d_oracle as O
    torch.manual_seed(0)
    W = torch.randn(8, 63, requires_grad=True)
    g = torch.Generator().manual_seed(1)
    x_all = torch.randn(64, 3, generator=g)
    # equal band sizes -> mean of per-band means == global mean
    (O.pos_embedding(x_all, 10) @ W.t()).pow(2).mean().backward()
    assert torch.allclose(res[0][:-5], W.grad.reshape(-1), rtol=1e-5, atol=1e-7)
    assert torch.allclose(res[0][-5:], torch.full((5,), 1.5))  # mean over ranks of d/db sum(b * (rank + 1)) = (1 + 2) / 2
