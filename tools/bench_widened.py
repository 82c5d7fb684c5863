"""Timing of the rows added after the per-sample path, which round 1 left untimed: skeleton FK
(one launch each way) against a torch loop over the joints, FlatAdamW against torch's foreach AdamW at the model's size, and
the hash-grid encoding's gather rate.  Run on the GPU box:  python tools/bench_widened.py  -> one JSON line per section."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEV = "cuda"


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def torch_fk_loop(local, so3, edges):
    """The reference's structure (skel_utils.py:50-94): exp map for all joints, then one (M,4,4) matmul per joint in a Python loop."""
    M, B = so3.shape[:2]
    th = so3.norm(dim=-1, keepdim=True).clamp_min(1e-6)
    v = so3 / th
    K = torch.zeros(M, B, 3, 3, device=so3.device)
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0], K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -v[..., 2], v[..., 1], v[..., 2], -v[..., 0], -v[..., 1], v[..., 0]
    th = th[..., None]
    R = torch.eye(3, device=so3.device) + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)
    loc = torch.eye(4, device=so3.device).expand(M, B, 4, 4).clone()
    loc[..., :3, :3], loc[..., :3, 3] = R, local
    glob = torch.eye(4, device=so3.device).expand(M, B, 4, 4).clone()
    for idx, par in edges.items():
        parent = glob[:, par - 1].clone() if par > 0 else torch.eye(4, device=so3.device).expand(M, 4, 4)
        glob[:, idx - 1] = parent @ loc[:, idx - 1]
    return glob


def bench_fk():
    from lab4d_amd import pose
    fx = torch.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pose.pt"), weights_only=False)
    edges, rest = fx["skel"]["edges"], fx["skel"]["rest_joints"].to(DEV)
    out = {}
    for M in (2, 256):
        so3 = (torch.randn(2 * M, 25, 3, device=DEV) * 0.5).requires_grad_(True)
        local = pose.rest_joints_to_local(rest, edges)[None].expand(2 * M, -1, -1).contiguous().requires_grad_(True)
        shift = torch.zeros(3, device=DEV, requires_grad=True)

        def hip():
            qr, qd = pose.fk_bones(local, so3, edges, shift=shift)
            (qr.sum() + qd.sum()).backward()

        def loop():
            torch_fk_loop(local, so3, edges).sum().backward()

        out["rows_%d" % (2 * M)] = {"hip_fwd_bwd_ms": round(timed(hip), 4), "torch_loop_fwd_bwd_ms": round(timed(loop, n=5), 4)}
    return out


def bench_optim():
    from lab4d_amd import optim, synthetic
    P = synthetic.make_weights(0)
    shapes = [v.shape for k, v in P.items() if v.dtype.is_floating_point and k != "aabb"]
    a = [torch.randn(s, device=DEV).requires_grad_(True) for s in shapes]
    b = [x.detach().clone().requires_grad_(True) for x in a]
    for p in a:
        p.grad = torch.randn_like(p)
    topt = torch.optim.AdamW(a, lr=5e-4, foreach=True)
    fopt = optim.FlatAdamW(b, 5e-4)
    fopt.flat_grad.normal_()

    def t_step():
        torch.nn.utils.clip_grad_norm_(a, 5.0)
        topt.step()

    n = sum(p.numel() for p in a)
    ms_f = timed(lambda: fopt.step(max_norm=5.0))
    return {"params": n, "tensors": len(a), "torch_clip_plus_foreach_adamw_ms": round(timed(t_step), 4), "flat_adamw_ms": round(ms_f, 4),
            "flat_adamw_GBps": round(n * 32 / (ms_f * 1e-3) / 1e9, 1)}


def bench_hashgrid():
    from lab4d_amd import hashgrid
    L, F, log2_T, S = 16, 2, 19, 1 << 22
    res = torch.tensor(hashgrid.level_resolutions(L, 16, 2048), dtype=torch.int32, device=DEV)
    table = (torch.randn(L, 1 << log2_T, F, device=DEV) * 0.1).requires_grad_(True)
    x = torch.rand(S, 3, device=DEV)
    fwd = timed(lambda: hashgrid.hash_encode(x, table, res, log2_T), n=10)
    out = hashgrid.hash_encode(x, table, res, log2_T)
    g = torch.randn_like(out)
    bwd = timed(lambda: torch.autograd.grad(hashgrid.hash_encode(x, table, res, log2_T), table, g), n=10) - fwd
    alg = S * (L * 8 * F * 4 + L * F * 4 + 12)
    return {"samples": S, "fwd_ms": round(fwd, 3), "bwd_ms": round(bwd, 3), "fwd_algorithmic_GBps": round(alg / (fwd * 1e-3) / 1e9, 1)}


if __name__ == "__main__":
    for name, fn in (("fk", bench_fk), ("optim", bench_optim), ("hashgrid", bench_hashgrid)):
        try:
            print(json.dumps({name: fn()}))
        except Exception as e:  # one broken section must not hide the others
            print(json.dumps({name: {"error": repr(e)[:300]}}))
