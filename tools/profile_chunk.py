"""torch.profiler view of ONE training chunk (eager): which torch ops (not lab4d kernels) cost device time, by op and shape.
usage: python tools/profile_chunk.py [chunk_rows]"""
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lab4d_amd import _lib, mlp  # noqa: E402
from lab4d_amd import deformable as DF  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32
res, spp = 512, 128
dev = torch.device("cuda", 0)
_lib.lib()
P, fr = bench.make_problem(res, dev)
from lab4d_amd.optim import FlatAdamW
opt = FlatAdamW([v for v in P.values() if v.dtype.is_floating_point and v.requires_grad], lr=5e-4)
mlp.FUSED_GRAD_ACCUM = True
_pro = DF.FramePrologue(P, fr)
fr = _pro.refresh()
hxy, batch = bench.chunk_inputs(res, None, list(range(0, res, res // rows)), dev, seed=100)
batch["hxy"] = hxy
gen = torch.Generator(device=dev).manual_seed(1)
M, N = hxy.shape[:2]
rng = bench.draw_rng(M, N, M * N * spp, dev, gen)
for _ in range(2):
    bench.train_chunk(DF, P, fr, hxy, batch, rng, spp, res, mlp.PREC_BF16)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    bench.train_chunk(DF, P, fr, hxy, batch, rng, spp, res, mlp.PREC_BF16)
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True)
rowsl = []
for e in ev:
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = getattr(e, "self_cuda_time_total", 0)
    if t <= 0:
        continue
    rowsl.append((t, e.count, e.key, str(e.input_shapes)[:90]))
rowsl.sort(reverse=True)
tot = sum(r[0] for r in rowsl)
print("total self device time %.2f ms over %d op groups" % (tot / 1e3, len(rowsl)))
for t, c, k, sh in rowsl[:70]:
    print("%9.1f us %5d x  %-38s %s" % (t, c, k[:38], sh))
byop = defaultdict(lambda: [0, 0])
for t, c, k, sh in rowsl:
    byop[k][0] += t
    byop[k][1] += c
print("\nby op:")
for k, (t, c) in sorted(byop.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%9.1f us %6d x  %s" % (t, c, k[:60]))
