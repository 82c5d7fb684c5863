"""Shared by the CPU (oracle) and GPU (device) parity tests and by tests/measure_fp32_noise_floor.py: how a reference-generated fixture's
weights and inputs are rebuilt from its seeds.  TEST INFRASTRUCTURE ONLY."""
import torch

from lab4d_amd import synthetic


def fg_weights(meta):
    """The fg field's weights of a fixture (tests/golden/make_golden.py: gen_train / gen_comp_train)."""
    motion = meta.get("fg_motion", "skel-quad")
    P = synthetic.make_weights(meta["seed"], num_inst=meta.get("num_inst", 1), sdf_bias=meta.get("sdf_bias"), num_bones=18 if "skel-human" in motion else 25,
                               motion=motion if motion in ("rigid", "dense") else "skinning")
    if motion.startswith("comp_"):
        P = synthetic.add_dense_weights(P, meta["seed"], meta.get("num_inst", 1))
    return P


def bg_weights(meta):
    Pb = synthetic.make_bg_weights(meta["seed"])
    Pb["sdf.bias"] = torch.tensor([meta["bg_sdf_bias"]])
    return Pb


def rays_and_targets(g):
    """(hxy, batch): stored by the small fixtures, regenerated from the seeds for the full-size ones (meta.full_grid_stride)."""
    meta = g["meta"]
    if "hxy" in g:
        return g["hxy"], g["batch"]
    hxy = synthetic.make_rays(meta["res"], meta["M"], rows=meta.get("rows"))
    return hxy, synthetic.make_targets(meta["seed"] + 3, meta["M"], hxy.shape[1], meta["res"], hxy)


def strided(g, t):
    """Every stride-th ray of a (M, N, ...) render, the way the full-size fixtures store it."""
    st = g["meta"].get("full_grid_stride")
    return t[:, ::st] if st and t.dim() >= 2 and t.shape[1] == g["meta"]["N"] else t


def leaf(P, dev=None, dt=None):
    out = {}
    for k, v in P.items():
        if dt is not None and v.dtype.is_floating_point:
            v = v.to(dt)
        if dev is not None:
            v = v.to(dev)
        out[k] = v.clone().requires_grad_(True) if v.dtype.is_floating_point and k != "aabb" else v
    return out
