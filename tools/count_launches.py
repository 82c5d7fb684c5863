"""Count device kernel launches per autograd-function / op name over one eager chunk (fwd+bwd)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lab4d_amd import deformable as DF, mlp
dev = torch.device("cuda")
P, fr = bench.make_problem(512, dev)
hxy, batch = bench.chunk_inputs(512, None, list(range(0, 512, 8)), dev, 1)  # one of the bench's 64-row chunks (rows spread over the frame)
from lab4d_amd.optim import FlatAdamW
opt = FlatAdamW([v for v in P.values() if v.dtype.is_floating_point and v.requires_grad], lr=5e-4)  # the bench's configuration:
mlp.FUSED_GRAD_ACCUM = True                                                                       # gradients accumulate in the flat buffer
gen = torch.Generator(device=dev).manual_seed(0)
M, N = hxy.shape[:2]
rng = bench.draw_rng(M, N, M * N * 128, dev, gen)
prologue = DF.FramePrologue(P, fr)   # as in bench.py: the per-frame terms are formed once per step, outside the chunk
fr = prologue.refresh()
for _ in range(2):
    bench.train_chunk(DF, P, fr, hxy, batch, rng, 128, 512, mlp.PREC_BF16)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    bench.train_chunk(DF, P, fr, hxy, batch, rng, 128, 512, mlp.PREC_BF16)
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = []
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        continue
    rows.append((e.key, e.count, e.self_device_time_total))
rows.sort(key=lambda r: -r[1])
tot = 0
print("top CPU-side ops by call count (aten ops launching kernels):")
for k, c, t in rows[:70]:
    print(f"{k[:60]:60s} {c:6d} {t/1e3:9.2f} ms self device")

# attribution: which line of the host layer issues the kernel-launching torch ops (aten ops with device time of their own)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof2:
    bench.train_chunk(DF, P, fr, hxy, batch, rng, 128, 512, mlp.PREC_BF16)
    torch.cuda.synchronize()
by_line = collections.Counter()
for e in prof2.key_averages(group_by_stack_n=12):
    if e.device_type == torch.autograd.DeviceType.CUDA or e.self_device_time_total <= 0 or not e.key.startswith("aten::"):
        continue
    where = next((f for f in e.stack if "lab4d_amd/" in f or "bench.py" in f), "(autograd engine / backward of torch ops)")
    by_line[where.split("/root/repo/")[-1][:110]] += e.count
print("\nkernel-launching aten ops by issuing line:")
for k, c in by_line.most_common(60):
    print(f"{c:5d}  {k}")
