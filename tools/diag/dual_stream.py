"""Experiment: do two half-chip pipelines beat one full-chip pipeline?  The chain kernels are CU-bound at ~3.1 TB/s and the weight-gradient kernels HBM-bound at
~5.7 TB/s with every CU resident; run on disjoint halves of the chip from two streams, one stream's weight gradients can overlap the other's chains.
    python tools/diag/dual_stream.py full            # one stream, S samples, full grids
    LAB4D_CHAIN_GRID=128 LAB4D_WGRAD_JOBS=128 python tools/diag/dual_stream.py dual   # two streams, S/2 samples each, half grids
(the grid overrides are read once per process, hence two invocations)"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lab4d_amd import _lib, mlp, synthetic

mode = sys.argv[1] if len(sys.argv) > 1 else "full"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 24
nets = (sys.argv[3] if len(sys.argv) > 3 else "base,color").split(",")
dev = "cuda"
P = synthetic.to_device(synthetic.make_weights(0), dev)
for k in P:
    if P[k].dtype.is_floating_point:
        P[k].requires_grad_(True)
        P[k].grad = torch.zeros_like(P[k])
mlp.FUSED_GRAD_ACCUM = True
fr = synthetic.to_device(synthetic.add_codes(synthetic.make_frames(1, 2, 512), synthetic.make_weights(0)), dev)
prec = mlp.PREC_BF16
nstream = 2 if mode == "dual" else 1
Ss = S // nstream
xs = [(torch.rand(Ss, 3, device=dev) * 0.3 - 0.15).requires_grad_(True) for _ in range(nstream)]
streams = [torch.cuda.Stream() for _ in range(nstream)]


def step(x):
    spf = x.shape[0] // 2
    sdf, feat = mlp.run_chain(mlp.NET_FG_BASE, prec, P, x, spf, conds={0: fr["code_base"], 4: fr["code_base"]}, export_layer=8)
    loss = sdf.sum()
    if "color" in nets:
        loss = loss + mlp.run_chain(mlp.NET_FG_COLOR, prec, P, x, spf, conds={0: fr["code_color"], 3: fr["appr_code"]}, ext=feat).sum()
    if "feat" in nets:
        loss = loss + mlp.run_chain(mlp.NET_FEAT, prec, P, x, spf).sum()
    loss.backward()


def run_all():
    for st, x in zip(streams, xs):
        with torch.cuda.stream(st):
            step(x)


for _ in range(2):
    run_all()
torch.cuda.synchronize()
n = 3
t0 = time.perf_counter()
for _ in range(n):
    run_all()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(json.dumps({"mode": mode, "S_total": S, "streams": nstream, "chain_grid": os.environ.get("LAB4D_CHAIN_GRID"), "wgrad_jobs": os.environ.get("LAB4D_WGRAD_JOBS"),
                  "nets": nets, "ms_per_pass": round(ms, 2)}))
