"""Device memory of one eager training chunk: allocated after the forward (= tensors saved for the backward), peak during the backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lab4d_amd import deformable as DF, mlp
from lab4d_amd.optim import FlatAdamW
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
res, spp = 512, 128
dev = torch.device("cuda", 0)
P, fr = bench.make_problem(res, dev)
opt = FlatAdamW([v for v in P.values() if v.dtype.is_floating_point and v.requires_grad], lr=5e-4)
mlp.FUSED_GRAD_ACCUM = True
pro = DF.FramePrologue(P, fr); fr = pro.refresh()
hxy, batch = bench.chunk_inputs(res, None, list(range(0, res, res // rows)), dev, seed=100)
batch["hxy"] = hxy
gen = torch.Generator(device=dev).manual_seed(1)
M, N = hxy.shape[:2]
S = M * N * spp
rng = bench.draw_rng(M, N, S, dev, gen)
G = 2**30
for it in range(2):
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    f = dict(fr); f["feature"] = batch["feature"]
    res_d = DF.render_train(P, f, hxy, rng, flow_thresh=float(res), n_depth=spp, prec=mlp.PREC_BF16)
    losses = DF.losses_fg(res_d, batch, res, DF.DEFAULT_LOSS_WT)
    torch.cuda.synchronize()
    after_fwd, peak_fwd = torch.cuda.memory_allocated(), torch.cuda.max_memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    losses.total.backward()
    torch.cuda.synchronize()
    peak_bwd = torch.cuda.max_memory_allocated()
    del res_d, losses
    print("iter %d  samples %.2f M | resident before %.1f GiB | saved by forward %.1f GiB = %.2f KB/sample (forward peak %.1f) | backward peak %.1f GiB = +%.2f KB/sample over the saved set"
          % (it, S / 1e6, base / G, (after_fwd - base) / G, (after_fwd - base) / S / 1024, (peak_fwd - base) / G, (peak_bwd - base) / G, (peak_bwd - after_fwd) / S / 1024))
