"""Capture ONE component of the comp training graph as a hipGraph at a small size and replay it (fault isolation).
usage: python tools/debug_graph.py {fg|bg|full|fgquad}"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lab4d_amd import deformable as DF, mlp, multifields, render_utils as RU
which = sys.argv[1]
dev = torch.device("cuda")
res, rows, spp = 128, 16, 128
if which == "fgquad":
    P, fr = bench.make_problem(res, dev)
    Pb = frb = None
else:
    P, fr, Pb, frb = bench.make_problem(res, dev, comp=True)
if len(sys.argv) > 2 and sys.argv[2] == "flat":
    from lab4d_amd.optim import FlatAdamW
    params = [v for k, v in P.items() if v.dtype.is_floating_point and v.requires_grad] + (list(Pb.values()) if Pb is not None else [])
    opt = FlatAdamW(params, lr=5e-4)
    mlp.FUSED_GRAD_ACCUM = True
hxy, batch = bench.chunk_inputs(res, 0, rows, dev, seed=3)
batch["hxy"] = hxy
gen = torch.Generator(device=dev).manual_seed(0)
M, N = hxy.shape[:2]
rng = bench.draw_rng(M, N, M * N * (spp // 2), dev, gen)
rng["eik_inds_bg"] = rng["eik_inds"]
pro = DF.FramePrologue(P, fr)
fr = pro.refresh(); pro.outs = None
if Pb is not None:
    prb = DF.BgPrologue(Pb, frb)
    frb = prb.refresh(); prb.outs = None


def run():
    f = dict(fr); f["feature"] = batch["feature"]
    if which in ("fg", "fgquad"):
        fd, d, aux = DF.query_field_train(P, f, hxy, rng, float(res), spp // 2, None, mlp.PREC_BF16)
        out = dict(RU.render_pixel(fd, d))
        tot = sum(v.sum() for v in out.values())
    elif which == "bg":
        fd, d, _ = DF.query_field_train_bg(Pb, frb, hxy, rng, float(res), spp // 2, None, mlp.PREC_BF16)
        out = dict(RU.render_pixel(fd, d))
        tot = sum(v.sum() for v in out.values())
    else:
        r = DF.render_train_comp(P, f, Pb, frb, hxy, rng, flow_thresh=float(res), n_depth=spp // 2, prec=mlp.PREC_BF16)
        tot = sum(DF.losses_comp(r, batch, res, DF.DEFAULT_LOSS_WT).values())
    tot.backward()
    return tot.detach()


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        print("eager total", float(run()), flush=True)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print(which, "eager ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    t = run()
torch.cuda.synchronize()
print(which, "captured", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print(which, "replayed ok", float(t), flush=True)
