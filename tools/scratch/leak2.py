import gc, sys, os, collections
sys.path.insert(0, "/root/repo")
import torch, bench
from lab4d_amd import deformable as DF, mlp
dev = torch.device("cuda")
P, fr = bench.make_problem(128, dev)
from lab4d_amd.optim import FlatAdamW
opt = FlatAdamW([v for v in P.values() if v.dtype.is_floating_point and v.requires_grad], lr=5e-4)
mlp.FUSED_GRAD_ACCUM = True
gen = torch.Generator(device=dev).manual_seed(0)
hxy, batch = bench.chunk_inputs(128, 32, 16, dev, 1)
M, N = hxy.shape[:2]
rng = bench.draw_rng(M, N, M * N * 32, dev, gen)
bench.train_chunk(DF, P, fr, hxy, batch, rng, 32, 128, mlp.PREC_BF16)
gc.collect()
gc.set_debug(gc.DEBUG_SAVEALL)
bench.train_chunk(DF, P, fr, hxy, batch, rng, 32, 128, mlp.PREC_BF16)
gc.collect()
c = collections.Counter(type(o).__name__ for o in gc.garbage)
print(c.most_common(40))
ids = {id(o) for o in gc.garbage}
for o in gc.garbage:
    n = type(o).__name__
    if "Backward" in n or n in ("LossDict",) or "Function" in n:
        refs = [type(r).__name__ for r in gc.get_referrers(o) if id(r) in ids]
        print(n, "<-", refs[:8])
