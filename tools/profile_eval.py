"""Where does eval-mode render_eval spend its time?  usage: python tools/profile_eval.py [rows]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lab4d_amd import _lib, mlp
from lab4d_amd import deformable as DF
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
_lib.lib()
P, fr = bench.make_problem(512, dev)
hxy, _ = bench.chunk_inputs(512, 0, rows, dev, seed=100)
for _ in range(2):
    DF.render_eval(P, fr, hxy, n_depth=128, prec=mlp.PREC_BF16)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    DF.render_eval(P, fr, hxy, n_depth=128, prec=mlp.PREC_BF16)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("render_eval: %.1f ms per call, %.0f rays/s" % (dt * 1e3, hxy.shape[0] * hxy.shape[1] / dt))
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    DF.render_eval(P, fr, hxy, n_depth=128, prec=mlp.PREC_BF16)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=22, max_name_column_width=60))
