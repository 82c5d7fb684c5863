import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lab4d_amd import deformable as DF, mlp, multifields, render_utils as RU
dev = torch.device("cuda")
res, rows, spp = 128, 16, 128
P, fr, Pb, frb = bench.make_problem(res, dev, comp=True)
hxy, batch = bench.chunk_inputs(res, 0, rows, dev, seed=3)
batch["hxy"] = hxy
gen = torch.Generator(device=dev).manual_seed(0)
M, N = hxy.shape[:2]
rng = bench.draw_rng(M, N, M * N * (spp // 2), dev, gen)
rng["eik_inds_bg"] = rng["eik_inds"]
f = dict(fr); f["feature"] = batch["feature"]
prec = mlp.PREC_BF16 if len(sys.argv) < 2 else mlp.PREC_F32
fd_fg, d_fg, aux = DF.query_field_train(P, f, hxy, rng, float(res), spp // 2, None, prec)
fd_bg, d_bg, _ = DF.query_field_train_bg(Pb, frb, hxy, rng, float(res), spp // 2, None, prec)
def rep(name, d):
    for k, v in d.items():
        if torch.is_tensor(v) and v.dtype.is_floating_point:
            n = int(torch.isnan(v).sum()); i = int(torch.isinf(v).sum())
            if n or i: print(name, k, tuple(v.shape), "nan", n, "inf", i)
rep("fg", fd_fg); rep("bg", fd_bg); rep("aux", aux)
fd, deltas = multifields.compose_fields({"fg": fd_fg, "bg": fd_bg}, {"fg": d_fg, "bg": d_bg})
rep("composed", fd)
r = dict(RU.render_pixel(fd, deltas)); rep("rendered", r)
rep("r_fg", dict(RU.render_pixel(fd_fg, d_fg))); rep("r_bg", dict(RU.render_pixel(fd_bg, d_bg)))
out = DF.render_train_comp(P, f, Pb, frb, hxy, rng, flow_thresh=float(res), n_depth=spp // 2, prec=prec)
L = DF.losses_comp(out, batch, res, DF.DEFAULT_LOSS_WT)
print({k: float(v) for k, v in L.items()})
