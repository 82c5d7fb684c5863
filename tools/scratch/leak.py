import gc, sys, os
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
def run(f, tag):
    for i in range(4):
        bench.train_chunk(DF, P, f, hxy, batch, rng, 32, 128, mlp.PREC_BF16)
        torch.cuda.synchronize()
        print(tag, i, "allocated MiB", torch.cuda.memory_allocated() >> 20, flush=True)
    n = gc.collect()
    print(tag, "after gc (collected %d)" % n, torch.cuda.memory_allocated() >> 20, flush=True)
run(fr, "inline")
pro = DF.FramePrologue(P, fr)
run(pro.refresh(), "prologue")
pro.backward()
print("after prologue.backward", torch.cuda.memory_allocated() >> 20)
