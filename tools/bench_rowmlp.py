"""What SURVEY 8f row 1 is about, measured: CameraMLP.get_vals-shaped work (TimeEmbedding + TimeMLP(D=5, W=256) + two heads, M = 256 frames) forward +
backward on the MI355X -- (a) as one rowmlp program (csrc/rowmlp.hip: 1 launch forward, 2 backward), (b) as the torch algebra of lab4d_amd/pose.py on
the same device (one launch per Linear / ReLU / cat / index, what the reference's modules do).  Wall time per forward + backward (the work is launch-bound:
0.17 GFLOP), the number of device kernels each path launches (torch profiler), the largest difference of the results.
usage: python tools/bench_rowmlp.py [M=256] [W=256]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lab4d_amd import pose  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda"
g = torch.Generator().manual_seed(0)
T, F = 512, 6
vid = torch.cat([torch.zeros(300, dtype=torch.long), torch.ones(T - 300, dtype=torch.long)])
info = {"frame_to_vid": vid, "frame_mapping": torch.arange(T), "raw_fid_to_vid": vid, "raw_fid_to_vidlen": torch.where(vid == 0, 300, T - 300),
        "raw_fid_to_vstart": torch.where(vid == 0, 0, 300), "max_ts": 300.0, "num_freq_t": F, "time_scale": 1.0}
info = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in info.items()}


def lin(o, i):
    return torch.randn(o, i, generator=g) * i ** -0.5, torch.randn(o, generator=g) * 0.1


P = {}
P["c.time_embedding.mapping1.weight"], P["c.time_embedding.mapping1.bias"] = lin(W, 2 * F + 1)
P["c.time_embedding.mapping2.weight"], P["c.time_embedding.mapping2.bias"] = lin(W, 2 * W)
P["c.time_embedding.inst_embedding.mapping.weight"] = torch.randn(2, W, generator=g)
for i in range(5):
    P[f"c.linear_{i+1}.0.weight"], P[f"c.linear_{i+1}.0.bias"] = lin(W, W)
P["c.linear_final.0.weight"], P["c.linear_final.0.bias"] = lin(W, W)
for head, o in (("trans", 3), ("quat", 4)):
    P[f"c.{head}.0.weight"], P[f"c.{head}.0.bias"] = lin(W // 2, W)
    P[f"c.{head}.2.weight"], P[f"c.{head}.2.bias"] = lin(o, W // 2)
P["c.base_quat"] = torch.randn(2, 4, generator=g)
P = {k: v.to(dev).requires_grad_(True) for k, v in P.items()}
fid = torch.randint(0, T, (M,), generator=g).to(dev)


def step():
    q, t = pose.camera_vals(P, "c", fid, info)
    (q.sum() + t.sum()).backward()
    return q, t


def measure(label):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
        step()
        torch.cuda.synchronize()
    kernels = sum(e.count for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA)
    return {"path": label, "ms_per_fwd_bwd": round(ms, 3), "device_kernels_per_fwd_bwd": int(kernels)}


a = measure("rowmlp program (csrc/rowmlp.hip), gradients returned to autograd")
qa, ta = [x.detach().clone() for x in step()]
# the patched Trainer's mode: every parameter's .grad is a view of the optimizer's flat buffer and the kernels add into it (no AccumulateGrad launch per parameter)
from lab4d_amd import mlp  # noqa: E402
from lab4d_amd.optim import FlatAdamW  # noqa: E402
opt = FlatAdamW(list(P.values()), lr=1e-3)
mlp.FUSED_GRAD_ACCUM = True
a2 = measure("rowmlp program, gradients accumulated into FlatAdamW's flat buffer by the kernels (the patched Trainer's mode)")
mlp.FUSED_GRAD_ACCUM = False
real = pose._on_gpu
pose._on_gpu = lambda P_, key: False  # the torch algebra on the same device tensors
try:
    b = measure("torch algebra, one launch per op (what the reference's modules do)")
    qb, tb = [x.detach().clone() for x in step()]
finally:
    pose._on_gpu = real
print(json.dumps({"M": M, "W": W, "paths": [a, a2, b], "max_abs_diff_quat": float((qa - qb).abs().max()), "max_abs_diff_trans": float((ta - tb).abs().max())}))
