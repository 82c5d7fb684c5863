"""CPU oracle for the multiresolution hash encoding (BASELINE config 5).

TEST INFRASTRUCTURE ONLY -- never imported by the product path (lab4d_amd/).

PARITY UNPINNED: the reference has no hash-grid encoding (lab4d/nnutils/nerf.py:98 is a TODO), so there are no
reference outputs, tests or golden vectors to pin this against.  It restates Mueller, Evans, Schied, Keller, "Instant Neural
Graphics Primitives with a Multiresolution Hash Encoding" (2022), section 3, in vectorised torch, independently of the
kernel-side header (different code structure: gather over all 8 corners at once, autograd for the adjoint):
  N_l = floor(N_min * b^l), b = exp((ln N_max - ln N_min) / (L - 1))                       (eq. 2, 3)
  vertices of the cell containing x * N_l; 1:1 indexing while (N_l + 1)^3 <= T, else
  h(v) = (v_x * 1  xor  v_y * 2654435761  xor  v_z * 805459861) mod T                     (eq. 4)
  tri-linear interpolation of the F-vectors, levels concatenated.
The FIELD on it (hash_field_forward) follows the paper's NeRF application (section 5.4 and appendix E: the hash grid is defined on the scene's
axis-aligned bounding box, mapped to [0,1]^3, and rays are marched INSIDE that box only): a sample outside the box carries no density and no
colour, and therefore no gradient to the table (round 4; rounds 1-3 clamped outside samples onto the boundary cells and evaluated them).
"""
import math

import torch

PRIMES = (1, 2654435761, 805459861)


def level_resolutions(L, n_min, n_max):
    b = math.exp((math.log(n_max) - math.log(n_min)) / (L - 1)) if L > 1 else 1.0
    return [int(math.floor(n_min * b ** l + 1e-9)) for l in range(L)]


def vertex_index(v, res, log2_T):
    """v: (..., 3) int64 vertex coordinates in [0, res]."""
    n, T = res + 1, 1 << log2_T
    if n ** 3 <= T:
        return v[..., 0] + n * (v[..., 1] + n * v[..., 2])
    h = (v[..., 0] * PRIMES[0]) ^ ((v[..., 1] * PRIMES[1]) & 0xFFFFFFFF) ^ ((v[..., 2] * PRIMES[2]) & 0xFFFFFFFF)
    return h & (T - 1)


def hash_encode(x, table, res_list, log2_T):
    """x (S,3) in [0,1]; table (L, T, F) -> (S, L*F).  Differentiable in x (piecewise) and table."""
    outs = []
    corners = torch.tensor([[c & 1, (c >> 1) & 1, c >> 2] for c in range(8)])
    for l, res in enumerate(res_list):
        p = x.clamp(0.0, 1.0) * float(res)
        i0 = torch.clamp(torch.floor(p.detach()), max=float(res - 1))
        w = p - i0                                                           # (S,3)
        v = i0.long()[:, None, :] + corners[None]                            # (S,8,3)
        idx = vertex_index(v, res, log2_T)                                   # (S,8)
        wc = torch.where(corners[None].bool(), w[:, None, :], 1 - w[:, None, :]).prod(-1)   # (S,8)
        outs.append((wc[..., None] * table[l][idx]).sum(1))
    return torch.cat(outs, -1)


def hash_field_forward(P, cfg, xyz, dirs, get_density=True):
    """Torch-CPU restatement of lab4d_amd/hashfield.forward (parity unpinned like the encoding itself): hash encoding -> geometry net
    32 -> 64 -> 16 (sdf = out[0]) -> VolSDF density (nerf.py:199-206's Laplace CDF) ; colour net [16 | dir] -> 64 -> 64 -> 3, sigmoid."""
    import torch.nn.functional as F
    lo, hi = P["aabb"][0], P["aabb"][1]
    x01 = (xyz - lo) / (hi - lo)
    enc = hash_encode(x01, P["hash.table"], level_resolutions(cfg["L"], cfg["n_min"], cfg["n_max"]), cfg["log2_T"])
    h = F.relu(F.linear(enc, P["hash.geo.0.weight"], P["hash.geo.0.bias"]))
    geo = F.linear(h, P["hash.geo.2.weight"], P["hash.geo.2.bias"])
    sdf = geo[:, :1]
    c = torch.cat([geo, dirs], -1)
    c = F.relu(F.linear(c, P["hash.color.0.weight"], P["hash.color.0.bias"]))
    c = F.relu(F.linear(c, P["hash.color.2.weight"], P["hash.color.2.bias"]))
    rgb = torch.sigmoid(F.linear(c, P["hash.color.4.weight"], P["hash.color.4.bias"]))
    inside = ((x01 >= 0) & (x01 <= 1)).all(-1, keepdim=True).to(rgb.dtype)  # Mueller et al. 2022, section 5.4 / appendix E: the grid lives on the box
    if not get_density:
        return rgb * inside, sdf
    ibeta = P["logibeta"].exp()
    density = (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() * ibeta)) * ibeta  # nerf.py:203-205
    return rgb * inside, density * inside
