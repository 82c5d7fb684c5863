"""Shared by the CPU (oracle) and GPU (device) parity tests and by tests/measure_fp32_noise_floor.py: how a reference-generated fixture's
weights and inputs are rebuilt from its seeds.  TEST INFRASTRUCTURE ONLY."""
import os

import torch

from lab4d_amd import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fg_weights(meta):
    """The fg field's weights of a fixture (tests/golden/make_golden.py: gen_train / gen_comp_train)."""
    motion = meta.get("fg_motion", "skel-quad")
    P = synthetic.make_weights(meta["seed"], num_inst=meta.get("num_inst", 1), sdf_bias=meta.get("sdf_bias"), num_bones=18 if "skel-human" in motion else 25,
                               motion=motion if motion in ("rigid", "dense") else "skinning")
    if motion.startswith("comp_"):
        P = synthetic.add_dense_weights(P, meta["seed"], meta.get("num_inst", 1))
    if meta.get("w1"):  # SURVEY 8d's fitted weight set: the tensors the reference's own geometry_init moved (tests/golden/make_golden.py: gen_w1_weights)
        w = torch.load(os.path.join(GOLDEN, "w1_weights.pt"), weights_only=False)
        assert w["seed"] == meta["seed"], (w["seed"], meta["seed"])
        P.update({k: v.clone() for k, v in w["changed"].items()})
    return P


def weight_checksum(P):
    return float(sum(v.double().abs().sum() for k, v in sorted(P.items()) if v.dtype.is_floating_point))


def eval_bench_unpack(g):
    """(inds (M, N, D/2) int64, valid (M, N, D) bool) of an eval_bench fixture (stored as uint8 / packed bits)."""
    import numpy as np
    shape = tuple(g["valid_shape"])
    n = shape[0] * shape[1] * shape[2]
    valid = torch.from_numpy(np.unpackbits(g["valid_bits"].numpy())[:n].astype(bool)).view(shape)
    return g["inds_u8"].long(), valid


def eval_bench_bands(g):
    """[(band index, hxy (M, n, 3), slice of the fixture's ray axis)] -- the fixture was rendered band by band (`band` image rows per call)."""
    meta = g["meta"]
    out = []
    for i, r0 in enumerate(range(meta["rows"][0], meta["rows"][1], meta["band"])):
        hxy = synthetic.make_rays(meta["res"], meta["M"], rows=(r0, r0 + meta["band"]))
        n = hxy.shape[1]
        out.append((i, hxy, slice(i * n, (i + 1) * n)))
    return out


def cdf_of_weights(weights, eps=1e-5):
    """The cdf render_utils.sample_pdf forms from (R, n) weights (render_utils.py:203-207), with torch's CPU ops -- the arithmetic the device kernel
    reproduces bit for bit (tests/test_sample_pdf_host.py, tests/test_gpu_ops.py).  weights: the coarse pass's (R, nc) compositing weights; the pdf uses
    the inner nc - 2 (nerf.py:721-727)."""
    w = weights[:, 1:-1].float().cpu() + eps
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    return torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)


def unpack_bits(bits, shape):
    import numpy as np
    n = 1
    for d in shape:
        n *= d
    return torch.from_numpy(np.unpackbits(bits.numpy())[:n].astype(bool)).view(tuple(shape))


def check_index_mismatches(g, band, inds_dev, cdf_dev, tol, ref=None, ties=None):
    """Every importance index that differs from the reference's must be a ONE-BIN shift at a cdf entry where the two implementations' cdfs straddle the
    query point and agree to `tol`: stored in the fixture as a near tie (the reference's cdf entries within meta.tie_window of a query point).
    inds_dev (M, n, D/2) int64 / cdf_dev (M, n, D/2 - 1 + ... ) of one band.  Returns (mismatches, worst |cdf_dev - cdf_ref| over them)."""
    meta = g["meta"]
    if ref is None:  # (comp fixtures pass their per-field index tensor and tie list)
        ref, _ = eval_bench_unpack(g)
    n = inds_dev.shape[1]
    ref = ref[:, band * n:(band + 1) * n]
    diff = (inds_dev != ref)
    if not bool(diff.any()):
        return 0, 0.0
    t = g["ties"] if ties is None else ties
    sel = t["band"] == band
    tie = {(int(m), int(nn), int(k)): float(c) for m, nn, k, c in zip(t["m"][sel], t["n"][sel], t["k"][sel], t["cdf"][sel])}
    u = g["u"]
    worst = 0.0
    for m, r, j in diff.nonzero().tolist():
        a, b = int(inds_dev[m, r, j]), int(ref[m, r, j])
        assert abs(a - b) == 1, ("index differs by more than one bin", band, m, r, j, a, b)
        k = min(a, b)  # searchsorted(right=True) counts the cdf entries <= u: the entry the two sides disagree about is entry min(a, b)
        assert (m, r, k) in tie, ("index differs at a cdf entry that is not within %g of a query point" % meta["tie_window"], band, m, r, j, a, b)
        c_ref, c_dev, uj = tie[(m, r, k)], float(cdf_dev[m, r, k]), float(u[j])
        lo, hi = min(c_ref, c_dev), max(c_ref, c_dev)
        assert lo <= uj <= hi or abs(c_ref - uj) <= tol or abs(c_dev - uj) <= tol, ("the two cdfs do not straddle the query point", c_ref, c_dev, uj)
        assert abs(c_ref - c_dev) <= tol, ("cdfs differ by more than the fp32 floor at a flipped index", c_ref, c_dev, uj)
        worst = max(worst, abs(c_ref - c_dev))
    return int(diff.sum()), worst


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
