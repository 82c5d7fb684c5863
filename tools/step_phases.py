"""Where the once-per-step work of the training loop goes (HIP events between the phases of TrainLoop.step, and the host time of the same phases):
usage: python tools/step_phases.py [rank_of=8]   (rank 0's rows of an N-rank job; 1 = the whole frame pair)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lab4d_amd import mlp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
res, spp = 512, 128
plan = bench.rank_plan(0, n, res, 128, spp)
loop = bench.TrainLoop(dev, res, spp, plan["chunks"], mlp.PREC_BF16)
for _ in range(3):
    loop.step()
torch.cuda.synchronize()
names = ["zero_grad+refresh", "chunks", "prologue.backward", "opt.step", "repack_all"]
acc = {k: [0.0, 0.0] for k in names}
steps = 8
for _ in range(steps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    t = [time.perf_counter()]
    ev[0].record()
    loop.opt.zero_grad(); loop.prologue.refresh()
    ev[1].record(); t.append(time.perf_counter())
    for hxy, batch in loop.inputs:
        loop.st_hxy.copy_(hxy)
        for k, v in batch.items():
            if k != "hxy":
                loop.st_batch[k].copy_(v)
        bench.draw_rng(loop.M, loop.N0, loop.S0, dev, loop.gen, out=loop.st_rng)
        loop.graph.replay()
    ev[2].record(); t.append(time.perf_counter())
    loop.prologue.backward()
    ev[3].record(); t.append(time.perf_counter())
    loop.opt.step(max_norm=5.0, skip_above=bench.GRAD_SKIP)
    ev[4].record(); t.append(time.perf_counter())
    mlp.repack_all()
    ev[5].record(); t.append(time.perf_counter())
    torch.cuda.synchronize()
    for i, k in enumerate(names):
        acc[k][0] += ev[i].elapsed_time(ev[i + 1])
        acc[k][1] += (t[i + 1] - t[i]) * 1e3
print(json.dumps({"rank_of": n, "ms_per_step": {k: {"device_span": round(v[0] / steps, 3), "host": round(v[1] / steps, 3)} for k, v in acc.items()}}))
