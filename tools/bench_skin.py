"""Micro-bench of the skinning warp (SkinningWarp.forward, warping.py:277-336) at the bench's launch size: forward + backward of ONE backward warp of
16,777,216 samples (one 128-row chunk of a 512x512 frame pair x 128 samples), the kernels of `kernels_ms_per_step`'s skinning family
(k_mlp_fwd<SkinA> inference, k_blend_fwd, k_blend_bwd+gram, k_mlp_bwd_fused<SkinA>) -- for rocprofv3 --pmc passes (tools/pmc_sq2.sh) and timing.
usage: python tools/bench_skin.py [S=16777216] [reps=3]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lab4d_amd import _lib, mlp  # noqa: E402
from lab4d_amd import deformable as DF  # noqa: E402
from lab4d_amd import warping as W  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16777216
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
_lib.lib()
P, fr = bench.make_problem(512, dev)
pro = DF.FramePrologue(P, fr)
fr = pro.refresh()
M = 2
g = torch.Generator(device=dev).manual_seed(3)
xyz = ((torch.rand(M, S // M // 128, 128, 3, device=dev, generator=g) - 0.5) * 0.3).requires_grad_(True)
w = torch.randn(M, S // M // 128, 128, 3, device=dev, generator=g)


def one():
    out, aux = W.skinning_warp(P, xyz, fr["t_articulation"], fr["rest_articulation"], fr["t_embed"], fr["code_skin"], True, mlp.PREC_BF16)
    loss = (out * w).sum() + 1e-3 * aux["skin_entropy"].sum() + 1e-3 * aux["delta_skin"].sum()
    loss.backward()


one()
torch.cuda.synchronize()
_lib.PROF = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    one()
e1.record()
torch.cuda.synchronize()
prof = _lib.prof_summary()
_lib.PROF = None
print(json.dumps({"S": S, "reps": reps, "ms_per_pass": round(e0.elapsed_time(e1) / reps, 3),
                  "kernels_ms_per_pass": {k: round(v[1] / reps, 3) for k, v in sorted(prof.items())}}))
