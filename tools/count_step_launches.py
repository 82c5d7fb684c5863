"""Host-side launches of ONE optimizer step of bench.TrainLoop (the driver command's shape) in its default whole-step-graph mode and in the per-chunk-graph
mode: kernel launches issued eagerly, graph launches, memcpy calls (torch.profiler, CPU activities).   python tools/count_step_launches.py"""
import collections
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lab4d_amd import _lib, mlp  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev = torch.device("cuda", 0)
_lib.lib()
out = {}
for mode, kw in (("step_graph", {"step_graph": True}), ("chunk_graph", {"step_graph": False})):
    mlp.clear_caches()
    loop = bench.TrainLoop(dev, 512, 128, bench.rank_plan(0, 1, 512, 128, 128)["chunks"], mlp.PREC_BF16, **kw)
    for _ in range(2):
        loop.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as pr:
        loop.step()
        torch.cuda.synchronize()
    c = collections.Counter()
    for e in pr.key_averages():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            continue
        k = e.key
        if "GraphLaunch" in k:
            c["graph_launches"] += e.count
        elif "LaunchKernel" in k or "launchkernel" in k.lower():
            c["eager_kernel_launches"] += e.count
        elif "emcpy" in k:
            c["memcpy_calls"] += e.count
    out[mode] = dict(c)
    out[mode]["runtime_calls_seen"] = sorted({e.key for e in pr.key_averages() if e.key.startswith(("hip", "cuda"))})[:12]
    loop.release_graph()
    del loop
    torch.cuda.empty_cache()
print(json.dumps(out))
