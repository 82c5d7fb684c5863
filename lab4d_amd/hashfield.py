"""A field on the multiresolution hash encoding (BASELINE config 5, "hash-grid (instant-NGP) encoding variant").

The reference has NO such field (lab4d/nnutils/nerf.py:98 is a TODO), so this mirrors no reference interface and its
parity against the reference is UNPINNED by nature; the definition follows Mueller et al. 2022 (sections 3 and 5.4) and is checked
against the independent restatement in oracle/hashgrid_oracle.py.  It is shaped like `NeRF.forward(xyz, dir, ..., get_density)`
(nnutils/nerf.py:167-215) so that it can stand where the positional-encoding field stands:
    x in aabb -> [0,1]^3 -> hash_encode (L levels x F features)        csrc/hashgrid.hip        (x outside the box: density = colour = 0)
      -> geometry net  L*F -> 64 -> 16: sdf = out[0], 15 geometry features     LAB4D_NET_HASH_GEO   (the fused chain kernels)
      -> density = VolSDF(sdf)                                                   csrc/flow.hip (as for the posenc field)
      -> colour net  [16 | view direction] -> 64 -> 64 -> 3, sigmoid            LAB4D_NET_HASH_COLOR
Parameters live in a flat dict like every other net here: "hash.table" (L, 2^log2_T, F), "hash.geo.{0,2}.{weight,bias}",
"hash.color.{0,2,4}.{weight,bias}", "logibeta".
"""
import math

import torch

from . import hashgrid, mlp
from .deformable import volsdf_density

DEFAULT = {"L": 16, "F": 2, "log2_T": 19, "n_min": 16, "n_max": 2048}


def make_weights(seed=0, cfg=None, sdf_bias=None):
    """Random-init parameters of the hash field (tables U(-1e-4, 1e-4) as in the paper, Linear layers Kaiming-uniform like nn.Linear)."""
    c = dict(DEFAULT, **(cfg or {}))
    g = torch.Generator().manual_seed(seed + 977)
    P = {"hash.table": (torch.rand(c["L"], 1 << c["log2_T"], c["F"], generator=g) * 2 - 1) * 1e-4}
    for name, o, i in [("hash.geo.0", 64, c["L"] * c["F"]), ("hash.geo.2", 16, 64), ("hash.color.0", 64, 19), ("hash.color.2", 64, 64), ("hash.color.4", 3, 64)]:
        b = 1.0 / math.sqrt(i)
        P[name + ".weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * b
        P[name + ".bias"] = (torch.rand(o, generator=g) * 2 - 1) * b
    if sdf_bias is not None:
        P["hash.geo.2.bias"][0] = float(sdf_bias)
    P["logibeta"] = torch.tensor([-math.log(0.1)])
    P["aabb"] = torch.tensor([[-0.12, -0.12, -0.12], [0.12, 0.12, 0.12]])
    return P, c


def resolutions(cfg, device):
    return torch.tensor(hashgrid.level_resolutions(cfg["L"], cfg["n_min"], cfg["n_max"]), dtype=torch.int32, device=device)


def forward(P, cfg, xyz, dirs, spf, prec=mlp.PREC_F32, res=None, get_density=True):
    """xyz, dirs (S,3) -> rgb (S,3), density or sdf (S,1).  spf = samples per frame (the chain kernels' frame stride; the hash nets have
    no per-frame conditioning, any positive value does)."""
    if cfg["L"] * cfg["F"] != 32:
        raise NotImplementedError("hash field: the geometry net is instantiated for L*F = 32 hash features")
    lo, hi = P["aabb"][0], P["aabb"][1]
    x01 = (xyz - lo) / (hi - lo)
    if res is None:
        res = resolutions(cfg, xyz.device)
    enc = hashgrid.hash_encode(x01, P["hash.table"], res, cfg["log2_T"], inside_only=True)  # (S, 32); zero rows outside the box (masked below)
    geo = mlp.run_chain(mlp.NET_HASH_GEO, prec, P, enc, spf)                       # (S, 16)
    sdf = geo[:, :1]
    rgb = torch.sigmoid(mlp.run_chain(mlp.NET_HASH_COLOR, prec, P, torch.cat([geo, dirs], -1), spf))
    # The grid is defined on the box (Mueller et al. 2022, section 5.4 / appendix E: rays are marched inside the scene's bounding box only): a sample
    # outside it has no density and no colour.  Its gradient towards the encoding is then exactly zero, and the table-gradient kernel issues no
    # atomic for a zero update (hashgrid_math.hpp: wave_run_add) -- in rounds 1-3 the 89 % of the bench's samples that lie outside the box were
    # clamped onto its boundary cells and their atomics were 68 % of that configuration's step.
    inside = ((x01 >= 0) & (x01 <= 1)).all(-1, keepdim=True).to(rgb.dtype)
    return rgb * inside, ((volsdf_density(sdf, P["logibeta"]) * inside) if get_density else sdf)
