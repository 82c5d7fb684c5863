import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lab4d_amd import deformable as DF, mlp
dev = torch.device("cuda")
res, spp = 512, 128
P, fr = bench.make_problem(res, dev)
pro = DF.FramePrologue(P, fr); fr = pro.refresh()
gen = torch.Generator(device=dev).manual_seed(0)
for row0 in (0, 64, 128, 224):
    hxy, batch = bench.chunk_inputs(res, row0, 8, dev, seed=3)
    batch["hxy"] = hxy
    M, N = hxy.shape[:2]
    rng = bench.draw_rng(M, N, M * N * spp, dev, gen)
    f = dict(fr); f["feature"] = batch["feature"]
    r = DF.render_train(P, f, hxy, rng, flow_thresh=float(res), n_depth=spp, prec=mlp.PREC_BF16)
    L = DF.losses_fg(r, batch, res, DF.DEFAULT_LOSS_WT)
    print(row0, "mask px", int(batch["mask"].sum()), {k: round(float(v), 5) for k, v in L.items()}, "total", float(L.total))
