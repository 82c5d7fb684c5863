"""Per-level cost of the hash-grid table gradient at the hash bench's chunk shape (1024^2 x 256 samples/ray, 16-row chunks: 8.4 M samples):
lab4d_hashgrid_backward called with ONE level at a time (L = 1) on the samples of a real chunk.  usage: python tools/bench_hashgrid_levels.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lab4d_amd import _lib, hashfield, hashgrid, synthetic
from lab4d_amd import quat_utils as Q, render_utils as RU

dev = torch.device("cuda", 0)
_lib.lib()
res, spp, rows = 1024, 256, 16
P, cfg = hashfield.make_weights(0, sdf_bias=0.02)
P = synthetic.to_device(P, dev)
fr = synthetic.to_device(synthetic.make_frames(1, 2, res), dev)
cam2field = Q.quaternion_translation_inverse(fr["field2cam"][0], fr["field2cam"][1])
hxy = synthetic.make_rays(res, 2, rows=list(range(0, res, res // rows))).to(dev)
_, _, deltas, _, xyz, dirs = RU.ray_samples(hxy, fr["Kinv"], fr["near_far"], cam2field, n_depth=spp)
lo, hi = P["aabb"][0], P["aabb"][1]
x01 = ((xyz.reshape(-1, 3) - lo) / (hi - lo)).contiguous()
S = x01.shape[0]
inside = ((x01 >= 0) & (x01 <= 1)).all(-1).float().mean().item()
levels = hashgrid.level_resolutions(cfg["L"], cfg["n_min"], cfg["n_max"])
out = {"S": S, "inside_fraction": round(inside, 4), "levels": []}
F, log2_T = cfg["F"], cfg["log2_T"]
g = torch.randn(S, F, device=dev)
for l, r in enumerate(levels):
    table = P["hash.table"][l:l + 1].contiguous()
    rt = torch.tensor([r], dtype=torch.int32, device=dev)
    gt = torch.zeros_like(table)
    gx = torch.empty_like(x01)
    ms = {}
    for what, a_gt, a_gx in (("table+x", gt, gx), ("x only", None, gx), ("table only", gt, None)):
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(_lib.lib().lab4d_hashgrid_backward(_lib.ptr(x01), _lib.ptr(table), _lib.ptr(rt), _lib.ptr(g), S, 1, log2_T, F, _lib.ptr(a_gt), _lib.ptr(a_gx),
                                                          _lib.stream()), "hashgrid_backward")
            e1.record()
            torch.cuda.synchronize()
            ms[what] = round(e0.elapsed_time(e1), 3)
    out["levels"].append({"level": l, "res": r, "dense": (r + 1) ** 3 <= (1 << log2_T), **ms})
print(json.dumps(out, indent=1))
