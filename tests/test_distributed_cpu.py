"""The N>1 path on CPU (gloo; world_size 2 and 8, one process per rank like the GPU job): bench.py's own round-robin row partition and
its ONE collective -- `bench.allreduce_flat` over the optimizer's flat gradient buffer (`lab4d_amd.optim.FlatAdamW.flat_grad`,
the product's bucket: parameters of different shapes, each padded to 4 elements) -- reproduce the data-parallel semantics of the
reference (DDP: every rank normalises its loss over its own rays, gradients are averaged over ranks), including an
uneven last band.  No scaling curve has been measured on hardware (no 8-GPU node was available to the builder); this is what
stands in for it, together with `bench.py --dry-ranks N`."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the sharding / all-reduce helpers of the real bench, not a copy)


def test_rank_rows_and_chunks_partition_the_frame():
    for res, world in [(512, 1), (512, 2), (512, 8), (500, 8), (64, 3)]:
        rows = [bench.rank_rows(r, world, res) for r in range(world)]
        assert sorted(y for rr in rows for y in rr) == list(range(res))
        assert max(len(rr) for rr in rows) - min(len(rr) for rr in rows) <= 1
        for rr in rows:  # a rank's rows dealt out to its chunks: a partition again, every chunk spread over the whole image
            for n in (1, 2, 4):
                ch = bench.chunk_rows_of(rr, n)
                assert sorted(y for c in ch for y in c) == rr
                assert all(c and max(c) - min(c) >= res - 2 * world * n for c in ch)


def test_dry_ranks_plans():
    """`bench.py --dry-ranks N`: every rank's band / chunk list / memory estimate; bands tile the frame, rays add up."""
    for world, res in [(8, 512), (4, 512), (2, 512), (3, 510)]:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-ranks", str(world), "--res", str(res)], capture_output=True, text=True,
                             timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads(out.stdout.strip().splitlines()[-1])
        assert d["world"] == world and len(d["plans"]) == world
        assert sum(p["rays_per_step"] for p in d["plans"]) == 2 * res * res
        if res == 512:
            assert all(p["uniform"] and p["est_peak_hbm_gib"] <= 200.0 and sum(p["chunk_sizes"]) == p["n_rows"] for p in d["plans"])


def test_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus N` with no launcher in front starts N ranks (round 3's flag was parsed and ignored).  On this GPU-less box: the
    RCCL form refuses loudly (non-zero exit, "needs N GPUs, found K"), the gloo dry step spawns 4 processes, runs the flat-gradient all-reduce and
    reports the ranks the process group saw; under a launcher a disagreeing --gpus is an error, not a relabelled 1-GPU run."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    b = os.path.join(ROOT, "bench.py")
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, b, "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode != 0 and "needs 2 GPUs, found %d" % torch.cuda.device_count() in out.stderr and out.stdout.strip() == ""
    out = subprocess.run([sys.executable, b, "--gpus", "4", "--backend", "gloo", "--dry-step", "--steps", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, lines  # exactly one JSON line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["ranks_seen_by_process_group"] == 4 and d["allreduce_is_rank_mean"] and d["rays_per_step_all_ranks"] == 2 * 512 * 512
    out = subprocess.run([sys.executable, b, "--gpus", "2", "--backend", "gloo", "--dry-step"], capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="3", RANK="0"))
    assert out.returncode != 0 and "must agree" in out.stderr


def _params():
    torch.manual_seed(0)  # same weights on every rank
    return [torch.randn(8, 63, requires_grad=True), torch.randn(5, requires_grad=True), torch.randn(3, 7, requires_grad=True), torch.randn(1, requires_grad=True)]


def _rank_loss(params, rank, world, res):
    """A stand-in for one rank's training loss: its rows, normalised over ITS rows (per-rank normaliser)."""
    from oracle import lab4d_oracle as O
    W, b, V, c = params
    g = torch.Generator().manual_seed(1)
    x_all = torch.randn(res, 3, generator=g)
    e = O.pos_embedding(x_all[bench.rank_rows(rank, world, res)], 10)
    return (e @ W.t()).pow(2).mean() + (b * (rank + 1)).sum() + (e[:, :7] @ V.t()).abs().mean() * c.sum()


def _worker(rank, world, port, res, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from lab4d_amd.optim import FlatAdamW
    params = _params()
    opt = FlatAdamW(params, lr=1e-3)  # flat layout on CPU tensors: p.grad are views of opt.flat_grad
    opt.zero_grad()
    _rank_loss(params, rank, world, res).backward()
    assert all(p.grad.data_ptr() >= opt.flat_grad.data_ptr() for p in params), "autograd must accumulate into the flat buffer"
    bench.allreduce_flat(opt.flat_grad, world)
    # by value (numpy), not as shared-memory tensors: the parent may pick the item up after this process has exited
    q.put((rank, opt.flat_grad.numpy().copy(), [p.grad.numpy().copy() for p in params]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,res,port", [(2, 64, 29561), (8, 64, 29571), (8, 500, 29581)])
def test_flat_gradient_allreduce_is_the_mean_of_the_rank_gradients(world, res, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, res, q)) for r in range(world)]
    [p.start() for p in ps]
    got = {}
    for _ in range(world):
        r, flat, grads = q.get(timeout=600)
        got[r] = (torch.from_numpy(flat), [torch.from_numpy(g) for g in grads])
    [p.join(60) for p in ps]
    for r in range(1, world):
        assert torch.equal(got[r][0], got[0][0]), "every rank holds the same reduced bucket"
    # expected: mean over ranks of each rank's own gradient (uneven shares at res=500: 63 rows for the first four ranks, 62 for the others)
    expect = None
    for r in range(world):
        params = _params()
        gs = torch.autograd.grad(_rank_loss(params, r, world, res), params)
        expect = [g / world for g in gs] if expect is None else [e + g / world for e, g in zip(expect, gs)]
    for g, e in zip(got[0][1], expect):
        assert torch.allclose(g, e, rtol=1e-5, atol=1e-7)
    if res % world == 0:  # equal shares: mean of per-rank means == global mean (the single-process gradient)
        params = _params()
        W = params[0]
        from oracle import lab4d_oracle as O
        x_all = torch.randn(res, 3, generator=torch.Generator().manual_seed(1))
        (gW,) = torch.autograd.grad((O.pos_embedding(x_all, 10) @ W.t()).pow(2).mean(), [W])
        part = torch.autograd.grad(sum((O.pos_embedding(x_all[bench.rank_rows(r, world, res)], 10) @ W.t()).pow(2).mean() for r in range(world)) / world, [W])[0]
        assert torch.allclose(gW, part, rtol=1e-5, atol=1e-7)
    # padding slots of the bucket (parameters are padded to 4 elements) stay zero through the reduction
    flat = got[0][0]
    used = torch.zeros_like(flat, dtype=torch.bool)
    off = 0
    for p in _params():
        used[off:off + p.numel()] = True
        off += (p.numel() + 3) // 4 * 4
    assert float(flat[~used].abs().sum()) == 0.0


# ---- the reference's own Trainer under DistributedDataParallel with the patched optimizer ----------------------------------
class _FusedLinear(torch.autograd.Function):
    """The FUSED_GRAD_ACCUM contract of lab4d_amd.mlp (mlp.py `_grad_sink`): the weight gradient is ADDED into weight.grad by the kernel, autograd is
    handed None for the weight -- here with a CPU matmul standing in for the weight-gradient kernel."""

    @staticmethod
    def forward(ctx, x, W):
        ctx.save_for_backward(x, W)
        return x @ W.t()

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        W.grad.add_(g.t() @ x)
        return g @ W, None


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.chain = torch.nn.Linear(6, 5, bias=False)   # a chain-kernel weight: fused accumulation
        self.head = torch.nn.Linear(5, 3)                # a per-frame module: plain autograd
        self.register_buffer("aabb", torch.ones(2, 3))   # (DDP broadcasts buffers every forward)

    def forward(self, x):
        return self.head(torch.relu(_FusedLinear.apply(x, self.chain.weight)))


def _toy_data(rank, it):
    g = torch.Generator().manual_seed(100 * it + rank)
    return torch.randn(7 + rank, 6, generator=g), torch.randn(7 + rank, 3, generator=g)


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from torch.nn.parallel import DistributedDataParallel
    from lab4d_amd import patch
    from lab4d_amd.optim import TorchFlatAdamW
    model = DistributedDataParallel(_Toy(), find_unused_parameters=False)   # engine/trainer.py:108-113
    opt = torch.optim.AdamW([{"params": [p], "lr": 1e-3} for p in model.parameters()], betas=(0.9, 0.999), weight_decay=1e-4)  # one group per parameter (trainer.py:164-190)
    TorchFlatAdamW.adopt(opt)                                               # what patch.trainer_optimizer_init does ...
    assert patch.ddp_local_accumulation(model)                              # ... incl. switching DDP's own reduction off
    opt.zero_grad()
    out = []
    for it in range(3):   # several iterations: an unreduced DDP bucket would raise at the second forward
        x, y = _toy_data(rank, it)
        if it > 0 and rank == 1:
            model.module.aabb.fill_(7.0)  # a buffer that drifted on one rank ...
        loss = (model(x) - y).pow(2).mean()
        assert float(model.module.aabb.sum()) == 6.0, "... is overwritten by rank 0's at the next forward, as under the reference's DDP"
        loss.backward()
        assert patch.allreduce_flat_grad(opt) == world                      # what patch.trainer_check_grad does in front of the clip
        patch.ddp_keep_buffer_sync(model)                                   # ... and: buffers keep following rank 0 (checked below)
        out.append(opt.flat.flat_grad.numpy().copy())
        opt.zero_grad()
        assert all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in model.parameters())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_patched_trainer_semantics_under_ddp_with_fused_grad_accum():
    """A model wrapped in DistributedDataParallel the way the reference's Trainer wraps it, the optimizer adopted the way patch() adopts it, one
    parameter on the fused-accumulation contract (autograd sees None for it): three iterations run (no "finished reduction" error from DDP's
    reducer), every rank ends every iteration with the SAME flat gradient = the mean over the ranks of each rank's own gradient."""
    world, port = 2, 29591
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    got = dict(q.get(timeout=600) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for it in range(3):
        assert (got[0][it] == got[1][it]).all()
        expect = None
        for r in range(world):
            m = _Toy()
            for p in m.parameters():
                p.grad = torch.zeros_like(p)
            x, y = _toy_data(r, it)
            (m(x) - y).pow(2).mean().backward()
            gs = [p.grad / world for p in m.parameters()]
            expect = gs if expect is None else [e + g for e, g in zip(expect, gs)]
        flat, off = torch.from_numpy(got[0][it]), 0
        for e in expect:
            assert torch.allclose(flat[off:off + e.numel()].view_as(e), e, rtol=1e-5, atol=1e-7)
            off += (e.numel() + 3) // 4 * 4
