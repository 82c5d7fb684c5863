"""Generate golden vectors by running the REFERENCE's own Python on CPU.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
Writes tests/golden/*.pt (small, committed).  The -m "not gpu" tests check the oracle
(oracle/lab4d_oracle.py) against these files; the -m gpu tests check the HIP path against
the oracle.  Weights are NOT stored: they are regenerated from lab4d_amd.synthetic
.make_weights(seed) and loaded into the reference modules with load_state_dict, a
checksum is stored instead.
"""
import os
import sys
from functools import partial

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
OUT_DIR = os.environ.get("LAB4D_GOLDEN_OUT", HERE)  # tests/test_golden_generator.py regenerates into a temp dir

from oracle import ref_shim  # noqa: E402
from lab4d_amd import synthetic  # noqa: E402

torch.set_num_threads(8)
_REAL_MULTINOMIAL = torch.multinomial  # gen_train / gen_comp_train replace torch.multinomial by the fixture's injected draw


def weight_checksum(P):
    return float(sum(v.double().abs().sum() for k, v in sorted(P.items()) if v.dtype.is_floating_point))


def compress_grad(g):
    """Big gradients are stored as (norm, strided subsample) to keep the fixture small.  The stride is made coprime with the row length, so the
    ~1,024 stored entries walk through EVERY input column (round 4: numel // 1024 = 64 on a (256, 256) weight stored columns 0, 64, 128, 192 only,
    four inputs whose gradients happen to be small for basefield.linear_4 -- the subsample's relative error was 16x the tensor's)."""
    if g.numel() <= 4096:
        return {"full": g.clone()}
    import math
    stride, cols = g.numel() // 1024, g.shape[-1]
    while math.gcd(stride, cols) != 1:
        stride += 1
    return {"norm": g.double().norm().float(), "stride": stride, "sub": g.flatten()[::stride].clone()}


def build_reference_field(ns, P, num_inst=1, fg_motion="skel-quad"):
    torch.manual_seed(0)
    di = ref_shim.synthetic_data_info(64)
    if num_inst > 1:  # one video per instance: the skinning / time modules size their tables from frame_info (skinning.py:70-86)
        import numpy as np
        off = np.asarray([round(64 * i / num_inst) for i in range(num_inst + 1)])
        di["frame_info"]["frame_offset"] = off
        di["frame_info"]["frame_offset_raw"] = off.copy()
    f = ns.deformable.Deformable(fg_motion, di, num_freq_dir=-1, appr_channels=32, num_inst=num_inst, init_scale=0.2)
    f.category = "fg"
    sd = {k: v for k, v in P.items() if k in f.state_dict()}
    missing = [k for k in P if k not in f.state_dict() and k != "warp.skinning_model.symm_idx"]
    assert not missing, missing
    f.load_state_dict(sd, strict=False)
    if hasattr(f.warp, "skinning_model"):
        assert list(f.warp.skinning_model.symm_idx) == synthetic.SYMM_IDX[len(f.warp.skinning_model.symm_idx)]
    else:  # fg_motion "rigid" / "dense": every warp parameter of the module must have come from P
        assert all(("warp." + k) in P for k in f.warp.state_dict() if "time_embedding" not in k), [k for k in f.warp.state_dict()]
    return f


def frames_from_reference(f, fr):
    """Per-frame codes the reference's own per-frame modules produce for these frame ids."""
    with torch.no_grad():
        fid = fr["frame_id"]
        fr["appr_code"] = f.appr_embedding.get_vals(fid).clone()
        if not hasattr(f.warp, "skinning_model"):
            if hasattr(f.warp, "time_embedding"):  # fg_motion "dense": the DenseWarp's own TimeEmbedding (warping.py:119)
                fr["t_embed_dense"] = f.warp.time_embedding(fid).clone()
            return fr
        fr["t_embed"] = f.warp.skinning_model.time_embedding(fid).clone()
        fr["t_embed_mean"] = f.warp.skinning_model.time_embedding.get_mean_embedding("cpu").clone()
        if hasattr(f.warp, "post_warp"):  # ComposedWarp: the dense post-warp has its own TimeEmbedding (warping.py:119)
            fr["t_embed_dense"] = f.warp.post_warp.time_embedding(fid).clone()
    return fr


def samples_dict_of(fr, hxy, feature):
    return {
        "Kinv": fr["Kinv"], "field2cam": fr["field2cam"], "frame_id": fr["frame_id"], "inst_id": fr["inst_id"],
        "near_far": fr["near_far"], "hxy": hxy, "feature": feature,
        "t_articulation": fr["t_articulation"], "rest_articulation": fr["rest_articulation"],
    }


def leafify(fr, names):
    out = dict(fr)
    leaves = {}
    for n in names:
        v = fr[n]
        if isinstance(v, tuple):
            v = tuple(t.clone().requires_grad_(True) for t in v)
            for i, t in enumerate(v):
                leaves[f"{n}.{i}"] = t
        else:
            v = v.clone().requires_grad_(True)
            leaves[n] = v
        out[n] = v
    return out, leaves


def w1_weights():
    """The fitted part of the W1 weight set (gen_w1_weights below), as committed."""
    return torch.load(os.path.join(HERE, "w1_weights.pt"), weights_only=False)


def gen_w1_weights(ns, seed=61):
    """SURVEY 8d's second weight set: W0 = make_weights(seed) after the REFERENCE'S OWN NeRF.geometry_init (nerf.py:251-295) driven by the Deformable
    override of get_init_sdf_fn (deformable.py:95-117: for skel-* motions the Gaussian-bone SDF of warping.py:338-353, no pysdf) -- 500 Adam steps at
    lr 1e-3 on 256 random points per step: sdf fit (scale-aligned), visibility prior, eikonal term.  Only tensors that moved are stored (the basefield,
    the sdf head, the visibility net and their instance codes: ~590 k numbers); fixture_utils.fg_weights(meta) overlays them when meta["w1"] is set.
    The fit is a chaotic iteration (500 optimiser steps): it is stored, not regenerated, by the generator test."""
    torch.multinomial = _REAL_MULTINOMIAL  # compute_eikonal's own draw (nerf.py:438-439), not a fixture's injected one
    P = synthetic.make_weights(seed)
    f = build_reference_field(ns, P)
    f.train()
    before = {k: v.detach().clone() for k, v in f.state_dict().items()}
    torch.manual_seed(seed)  # geometry_init draws its points / instance ids from the global generator
    with torch.no_grad():
        pts = f.sample_points_aabb(4096, extend_factor=0.25)
        sdf_gt = f.get_init_sdf_fn()(pts)
        sdf0 = f.forward(pts, inst_id=None, get_density=False)
    f.geometry_init(f.get_init_sdf_fn())
    with torch.no_grad():
        sdf1 = f.forward(pts, inst_id=None, get_density=False)
    after = f.state_dict()
    changed = {k: after[k].detach().clone() for k in P if k in after and not torch.equal(after[k], before[k])}
    corr = lambda a, b: float(torch.corrcoef(torch.stack([a.flatten(), b.flatten()]))[0, 1])
    out = {"seed": seed, "changed": changed, "weight_checksum_w0": weight_checksum(P),
           "weight_checksum_w1": weight_checksum(dict(P, **changed)),
           "fit": {"corr_before": corr(sdf0, sdf_gt), "corr_after": corr(sdf1, sdf_gt), "inside_frac_gt": float((sdf_gt < 0).float().mean()),
                   "inside_frac_after": float((sdf1 < 0).float().mean())}}
    path = os.path.join(OUT_DIR, "w1_weights.pt")
    torch.save(out, path)
    print("w1_weights ->", path, os.path.getsize(path) // 1024, "KiB", sorted(changed)[:4], "...", len(changed), "tensors", out["fit"])


def gen_train(ns, tag, M, N, D, res, seed, alpha=None, num_inst=1, inst_id=None, frame_id=None, full_grid_stride=None, fg_motion="skel-quad",
              rows=None, w1=False):
    """num_inst > 1 / inst_id: the multi-instance configuration (BASELINE config 4): per-instance codes in every CondMLP
    (base.py:123-150), frames of one pair share their video's instance id."""
    num_bones = 18 if "skel-human" in fg_motion else 25  # utils/skel_utils.py:348-351
    P = synthetic.make_weights(seed, num_inst=num_inst, num_bones=num_bones, motion=fg_motion if fg_motion in ("rigid", "dense") else "skinning")
    if fg_motion.startswith("comp_"):  # fg_motion "comp_skel-quad_dense" (BASELINE configs 2-3): skinning + dense post-warp
        P = synthetic.add_dense_weights(P, seed, num_inst)
    if w1:  # the fitted weight set (gen_w1_weights): a sharp surface instead of the raw initialisation's flat field
        w = w1_weights()
        assert w["seed"] == seed and abs(w["weight_checksum_w0"] - weight_checksum(P)) <= 1e-6 * w["weight_checksum_w0"]
        P.update(w["changed"])
    f = build_reference_field(ns, P, num_inst, fg_motion)
    f.train()
    f.pos_embedding.set_alpha(alpha)
    f.pos_embedding_color.set_alpha(alpha)
    fr = synthetic.make_frames(seed + 1, M, res, num_bones=num_bones)
    if inst_id is not None:
        fr["inst_id"] = torch.tensor(inst_id, dtype=torch.long)
    if frame_id is not None:
        fr["frame_id"] = torch.tensor(frame_id, dtype=torch.long)
    fr = frames_from_reference(f, fr)
    g = torch.Generator().manual_seed(seed + 2)
    if full_grid_stride:  # BASELINE config 0: the whole res x res crop of every frame; only every stride-th ray is stored
        hxy = synthetic.make_rays(res, M, rows=rows)  # rows=(y0, y1): a band of image rows of every frame (the bench's chunk shape)
        N = hxy.shape[1]
    else:
        hxy = torch.cat([torch.rand(M, N, 2, generator=g) * res, torch.ones(M, N, 1)], -1)
    batch = synthetic.make_targets(seed + 3, M, N, res, hxy)
    eik_n = max(M * N // 16, 1)
    eik_inds = torch.randperm(M * N, generator=g)[:eik_n]
    match_perm = torch.randperm(M * N * D, generator=g)[: min(1024, M * N * D)]
    # inject host-side randomness (nerf.py:438-439 multinomial, feature.py:177 randperm)
    ns.nerf.torch.multinomial = lambda probs, n, replacement=False: eik_inds.clone()
    ns.feature.torch = ns.nerf.torch.__class__(**vars(ns.nerf.torch))
    ns.feature.torch.randperm = lambda n: torch.cat([match_perm, torch.arange(n)])  # [:num_candidates] is taken
    # D samples per ray (n_depth is reachable only through the kwarg)
    ns.nerf.sample_cam_rays = partial(ns.render_utils.sample_cam_rays, n_depth=D)

    fr_l, leaves = leafify(fr, ["Kinv", "field2cam", "t_articulation", "rest_articulation"])
    sd = samples_dict_of(fr_l, hxy, batch["feature"])
    feat_dict, deltas, aux = f.query_field(sd, flow_thresh=float(res))
    rendered = ns.render_utils.render_pixel(feat_dict, deltas)
    aux_fg = dict(aux)
    aux_fg.update(ns.render_utils.render_pixel(feat_dict, deltas))
    results = {"rendered": dict(rendered), "aux_dict": {"fg": aux_fg}}
    results["rendered"]["xyz_matches"] = aux["xyz_matches"]
    results["rendered"]["xyz_reproj"] = aux["xyz_reproj"]

    import importlib
    model = importlib.import_module("lab4d.engine.model").dvr_model
    config = {"field_type": "fg", "train_res": res}
    loss_dict = {}
    model.compute_recon_loss(loss_dict, results, batch, config)
    model.mask_losses(loss_dict, batch, config)
    if "gauss_mask" not in rendered:
        assert fg_motion in ("rigid", "dense")  # deformable.py:344: the gaussian-bone density exists for SkinningWarp only
    loss_dict["reg_eikonal"] = rendered["eikonal"]
    loss_dict["reg_deform_cyc"] = aux_fg["cyc_dist"]
    loss_dict["reg_delta_skin"] = aux_fg["delta_skin"]
    loss_dict["reg_skin_entropy"] = aux_fg["skin_entropy"]
    from oracle.lab4d_oracle import DEFAULT_LOSS_WT
    config.update(DEFAULT_LOSS_WT)
    model.apply_loss_weights(loss_dict, config)
    # a term without a positive element is the mean of an empty selection = NaN in the reference (model.py:602; with fg_motion "rigid" that
    # is reg_deform_cyc / reg_delta_skin / reg_skin_entropy, all identically zero): it contributes no gradient, and the sum the trainer
    # backpropagates (trainer.py:343-345) is NaN only as a number -- the gradients below are those of the finite terms
    total = sum(v for v in loss_dict.values() if bool(torch.isfinite(v)))
    params = dict(f.named_parameters())
    gnames = [k for k in P if k in params and params[k].requires_grad]
    grads = torch.autograd.grad(total, [params[k] for k in gnames] + list(leaves.values()), allow_unused=True)
    gd = {}
    for k, gv in zip(gnames + ["frame:" + k for k in leaves], grads):
        if gv is not None:
            gd[k] = compress_grad(gv.detach())
    out = {
        "meta": {"M": M, "N": N, "D": D, "res": res, "seed": seed, "alpha": alpha, "num_inst": num_inst, "fg_motion": fg_motion,
                 "weight_checksum": weight_checksum(P), "flow_thresh": float(res), **({"w1": True} if w1 else {})},
        "frames": {k: (tuple(t.detach() for t in v) if isinstance(v, tuple) else v.detach()) for k, v in fr.items()},
        "hxy": hxy, "batch": batch, "rng": {"eik_inds": eik_inds.clone(), "match_perm": match_perm.clone()},
        "feat_dict": {k: v.detach() for k, v in feat_dict.items()}, "deltas": deltas.detach(),
        "rendered": {k: v.detach() for k, v in results["rendered"].items()},
        "aux_fg": {k: v.detach() for k, v in aux_fg.items()},
        "loss": {k: v.detach() for k, v in loss_dict.items()}, "grads": gd,
    }
    if full_grid_stride:  # compact fixture: inputs are regenerated from the seeds by the tests (make_rays / make_targets)
        st = full_grid_stride
        out["meta"]["full_grid_stride"] = st
        out["meta"]["rows"] = rows
        for k in ("hxy", "batch", "feat_dict", "deltas"):
            out.pop(k)
        out["rendered"] = {k: v[:, ::st].clone() for k, v in out["rendered"].items()}
        out["aux_fg"] = {k: (v[:, ::st].clone() if v.dim() >= 2 and v.shape[1] == N else v) for k, v in out["aux_fg"].items()}
    path = os.path.join(OUT_DIR, f"train_{tag}.pt")
    torch.save(out, path)
    print(tag, "->", path, os.path.getsize(path) // 1024, "KiB", {k: float(v) for k, v in loss_dict.items()})


def gen_eval(ns, tag, M, N, D, res, seed, fg_motion="skel-quad"):
    P = synthetic.make_weights(seed, sdf_bias=-0.02, motion=fg_motion if fg_motion in ("rigid", "dense") else "skinning")
    f = build_reference_field(ns, P, fg_motion=fg_motion)
    f.eval()
    fr = synthetic.make_frames(seed + 1, M, res)
    fr = frames_from_reference(f, fr)
    g = torch.Generator().manual_seed(seed + 2)
    hxy = torch.cat([torch.rand(M, N, 2, generator=g) * res, torch.ones(M, N, 1)], -1)
    # n_depth for the eval branch (nerf.py:696)
    orig = f.importance_sampling
    f.importance_sampling = lambda *a, **k: orig(*a, n_depth=D, **k)
    sd = samples_dict_of(fr, hxy, None)
    del sd["feature"]
    if fg_motion in ("rigid", "dense"):  # only SkinningWarp fields put articulations into the samples (deformable.py:254-289)
        del sd["t_articulation"], sd["rest_articulation"]
    captured = {}
    _sp = ns.render_utils.sample_pdf
    _ss = torch.searchsorted

    def searchsorted(cdf, u, right=False):
        r = _ss(cdf, u, right=right)
        captured["inds"] = r.clone()
        return r

    ns.render_utils.torch.searchsorted = searchsorted
    _gv = f.get_valid_idx

    def get_valid_idx(*a, **k):
        v = _gv(*a, **k)
        captured["valid"] = v.clone()
        return v

    f.get_valid_idx = get_valid_idx
    try:
        feat_dict, deltas, aux = f.query_field(sd)
    finally:
        ns.render_utils.torch.searchsorted = _ss
    rendered = ns.render_utils.render_pixel(feat_dict, deltas)
    out = {
        "meta": {"M": M, "N": N, "D": D, "res": res, "seed": seed, "weight_checksum": weight_checksum(P), "sdf_bias": -0.02, "fg_motion": fg_motion},
        "frames": {k: (tuple(t.detach() for t in v) if isinstance(v, tuple) else v.detach()) for k, v in fr.items()},
        "hxy": hxy, "feat_dict": {k: v.detach() for k, v in feat_dict.items()}, "deltas": deltas.detach(),
        "rendered": {k: v.detach() for k, v in rendered.items()},
        "inds": captured["inds"], "valid": captured["valid"],
    }
    path = os.path.join(OUT_DIR, f"eval_{tag}.pt")
    torch.save(out, path)
    print(tag, "->", path, os.path.getsize(path) // 1024, "KiB", "valid frac", float(captured["valid"].float().mean()))


def gen_eval_bench(ns, tag, res=512, D=128, seed=61, rows=(252, 260), band=2, stride=16, w1=False, tie_window=1e-4):
    """The eval path (render.py:183 -> dvr_model.evaluate -> NeRF.query_field in eval mode) at BASELINE configs[1]'s size: `rows` image rows of a
    512x512 frame pair, importance sampling with n_depth = 128 (64 uniform + 64 inverse-cdf samples, nerf.py:686-738), compute_normal on every
    sample (nerf.py:455-493), get_valid_idx + query_nerf compaction (nerf.py:495-528, 769-819) -- run through the reference band by band
    (`band` rows = 2 x 1,024 rays per call: the reference itself renders big inputs chunk by chunk, engine/model.py:259-326; rays are independent
    except for the mean-transmittance normaliser of `vis`, which is per call).  Stored for EVERY ray: the importance indices (uint8) and the valid
    mask (bit-packed); every `stride`-th ray of the render; and the reference's cdf at its NEAR TIES -- every cdf entry within `tie_window` of one
    of the 64 query points u: an index can differ between two fp32 implementations only at such an entry, and the device test asserts exactly that."""
    import numpy as np
    M = 2
    P = synthetic.make_weights(seed, sdf_bias=None if w1 else -0.02)
    if w1:
        w = w1_weights()
        assert w["seed"] == seed
        P.update(w["changed"])
    f = build_reference_field(ns, P)
    f.eval()
    fr = synthetic.make_frames(seed + 1, M, res)
    fr = frames_from_reference(f, fr)
    orig = f.importance_sampling
    f.importance_sampling = lambda *a, **k: orig(*a, n_depth=D, **k)
    captured = {}
    _ss = torch.searchsorted

    def searchsorted(cdf, u, right=False):
        r = _ss(cdf, u, right=right)
        captured["inds"], captured["cdf"], captured["u"] = r.clone(), cdf.clone(), u.clone()
        return r

    _gv = f.get_valid_idx

    def get_valid_idx(*a, **k):
        v = _gv(*a, **k)
        captured["valid"] = v.clone()
        return v

    f.get_valid_idx = get_valid_idx
    inds, valid, rendered, ties = [], [], [], []
    ns.render_utils.torch.searchsorted = searchsorted
    try:
        for r0 in range(rows[0], rows[1], band):
            hxy = synthetic.make_rays(res, M, rows=(r0, r0 + band))
            sd = samples_dict_of(fr, hxy, None)
            del sd["feature"]
            feat_dict, deltas, aux = f.query_field(sd)
            out = ns.render_utils.render_pixel(feat_dict, deltas)
            inds.append(captured["inds"].view(M, -1, D // 2))
            valid.append(captured["valid"].view(M, -1, D))
            rendered.append({k: v.detach()[:, ::stride].clone() for k, v in out.items()})
            cdf, u = captured["cdf"].view(M, -1, D // 2 - 1), captured["u"][0]
            assert torch.equal(captured["u"], u[None].expand_as(captured["u"]))
            near = (cdf[..., None] - u).abs().min(-1)[0] < tie_window
            m, n, k = near.nonzero(as_tuple=True)
            ties.append({"band": torch.full_like(m, (r0 - rows[0]) // band), "m": m, "n": n, "k": k, "cdf": cdf[m, n, k].clone()})
            print("  band", r0, "valid frac %.3f" % float(captured["valid"].float().mean()), "near ties", int(near.sum()), flush=True)
    finally:
        ns.render_utils.torch.searchsorted = _ss
    inds = torch.cat(inds, 1)    # (M, N, D/2): bands are consecutive rows, so this is the ray order of make_rays(res, M, rows)
    valid = torch.cat(valid, 1)  # (M, N, D)
    assert int(inds.max()) < 256
    out = {
        "meta": {"M": M, "N": inds.shape[1], "D": D, "res": res, "seed": seed, "rows": rows, "band": band, "full_grid_stride": stride,
                 "weight_checksum": weight_checksum(P), "sdf_bias": None if w1 else -0.02, "fg_motion": "skel-quad", "w1": bool(w1), "tie_window": tie_window},
        "frames": {k: (tuple(t.detach() for t in v) if isinstance(v, tuple) else v.detach()) for k, v in fr.items()},
        "inds_u8": inds.to(torch.uint8), "valid_bits": torch.from_numpy(np.packbits(valid.numpy().reshape(-1))), "valid_shape": tuple(valid.shape),
        "u": u.clone(),
        # per band: the renders are per call (the `vis` channel is normalised by the call's mean transmittance, render_utils.py:89)
        "rendered_bands": rendered,
        "ties": {k: torch.cat([t[k] for t in ties]).to(torch.int32 if k != "cdf" else torch.float32) for k in ties[0]},
    }
    path = os.path.join(OUT_DIR, f"eval_{tag}.pt")
    torch.save(out, path)
    print(tag, "->", path, os.path.getsize(path) // 1024, "KiB", "valid frac", float(valid.float().mean()), "indices", inds.numel(), "samples", valid.numel(),
          "near ties", out["ties"]["k"].numel(), "mask mean", float(torch.cat([r["mask"] for r in rendered], 1).mean()))


def gen_comp_eval_bench(ns, tag="comp_eval_bench", res=512, D=64, seed=131, rows=(254, 258), band=2, stride=16, fg_motion="comp_skel-human_dense", frame_id=(10, 11),
                        tie_window=1e-4):
    """field_type "comp" in eval mode at BASELINE configs[2]'s per-GPU shape (round 5): fg Deformable(comp_skel-human_dense: 18 bones + dense post-warp) +
    bg NeRF on the same rays, each with importance sampling n_depth = 64 (32 + 32), normals, the fg's valid-index compaction, then compose_fields
    (z-merge to 128 samples) and render_pixel of the composite and of each field -- 4 image rows of a 512x512 pair = 4,096 rays, band by band like
    gen_eval_bench.  Stored: both fields' importance indices (uint8) and the fg valid mask (bits) for EVERY ray, the near ties of both cdfs, every
    stride-th ray of the three renders."""
    import numpy as np
    M = 2
    num_bones = 18 if "skel-human" in fg_motion else 25
    Pf = synthetic.make_weights(seed, sdf_bias=-0.02, num_bones=num_bones)
    if fg_motion.startswith("comp_"):
        Pf = synthetic.add_dense_weights(Pf, seed, 1)
    f = build_reference_field(ns, Pf, 1, fg_motion)
    f.eval()
    fr0 = synthetic.make_frames(seed + 1, M, res, num_bones=num_bones)
    fr0["frame_id"] = torch.tensor(frame_id, dtype=torch.long)
    frf = frames_from_reference(f, fr0)
    Pb = synthetic.make_bg_weights(seed)
    Pb["sdf.bias"] = torch.tensor([-0.1])
    torch.manual_seed(0)
    di = ref_shim.synthetic_data_info(64)
    b = ns.nerf.NeRF(di, num_freq_xyz=6, num_freq_dir=0, appr_channels=0, init_scale=0.1)
    b.category = "bg"
    b.load_state_dict({k: v for k, v in Pb.items()}, strict=False)
    b.eval()
    frb = synthetic.make_bg_frames(seed + 3, M, res)
    for fld in (f, b):
        orig = fld.importance_sampling
        fld.importance_sampling = (lambda o: (lambda *a, **k: o(*a, n_depth=D, **k)))(orig)
    captured = {}
    _ss = torch.searchsorted

    def searchsorted(cdf, u, right=False):
        r = _ss(cdf, u, right=right)
        captured.setdefault("ss", []).append((r.clone(), cdf.clone(), u.clone()))
        return r

    _gv = f.get_valid_idx

    def get_valid_idx(*a, **k):
        v = _gv(*a, **k)
        captured["valid"] = v.clone()
        return v

    f.get_valid_idx = get_valid_idx
    acc = {"inds_fg": [], "inds_bg": [], "valid": [], "rendered": [], "rendered_fg": [], "rendered_bg": [], "ties_fg": [], "ties_bg": []}
    ns.render_utils.torch.searchsorted = searchsorted
    det = lambda d: {k: v.detach()[:, ::stride].clone() for k, v in d.items()}  # noqa: E731
    try:
        for bi, r0 in enumerate(range(rows[0], rows[1], band)):
            hxy = synthetic.make_rays(res, M, rows=(r0, r0 + band))
            sdf_ = samples_dict_of(frf, hxy, None)
            del sdf_["feature"]
            sdb = {"Kinv": frb["Kinv"], "field2cam": frb["field2cam"], "frame_id": frb["frame_id"], "inst_id": frb["inst_id"], "near_far": frb["near_far"], "hxy": hxy}
            captured["ss"] = []
            fd_f, d_f, _ = f.query_field(sdf_)
            fd_b, d_b, _ = b.query_field(sdb)
            assert len(captured["ss"]) == 2  # one sample_pdf per field, fg first
            comp, dcomp = ns.multifields.MultiFields.compose_fields({"fg": dict(fd_f), "bg": dict(fd_b)}, {"fg": d_f, "bg": d_b})
            acc["rendered"].append(det(ns.render_utils.render_pixel(comp, dcomp)))
            acc["rendered_fg"].append(det(ns.render_utils.render_pixel(fd_f, d_f)))
            acc["rendered_bg"].append(det(ns.render_utils.render_pixel(fd_b, d_b)))
            acc["valid"].append(captured["valid"].view(M, -1, D))
            for name, (inds, cdf, u) in zip(("fg", "bg"), captured["ss"]):
                acc["inds_" + name].append(inds.view(M, -1, D // 2))
                cdf, u0 = cdf.view(M, -1, D // 2 - 1), u[0]
                near = (cdf[..., None] - u0).abs().min(-1)[0] < tie_window
                m, n, k = near.nonzero(as_tuple=True)
                acc["ties_" + name].append({"band": torch.full_like(m, bi), "m": m, "n": n, "k": k, "cdf": cdf[m, n, k].clone()})
            print("  band", r0, "fg valid frac %.3f" % float(captured["valid"].float().mean()), flush=True)
    finally:
        ns.render_utils.torch.searchsorted = _ss
    cat = lambda xs: torch.cat(xs, 1)  # noqa: E731
    valid = cat(acc["valid"])
    ties = lambda ts: {k: torch.cat([t[k] for t in ts]).to(torch.int32 if k != "cdf" else torch.float32) for k in ts[0]}  # noqa: E731
    out = {"meta": {"M": M, "N": valid.shape[1], "D": D, "res": res, "seed": seed, "rows": rows, "band": band, "full_grid_stride": stride, "fg_motion": fg_motion,
                    "bg_sdf_bias": -0.1, "sdf_bias": -0.02, "tie_window": tie_window, "weight_checksum_fg": weight_checksum(Pf), "weight_checksum_bg": weight_checksum(Pb)},
           "frames_fg": {k: (tuple(t.detach() for t in v) if isinstance(v, tuple) else v.detach()) for k, v in frf.items()},
           "frames_bg": {k: (tuple(t.detach() for t in v) if isinstance(v, tuple) else v.detach()) for k, v in frb.items()},
           "inds_fg_u8": cat(acc["inds_fg"]).to(torch.uint8), "inds_bg_u8": cat(acc["inds_bg"]).to(torch.uint8),
           "valid_bits": torch.from_numpy(np.packbits(valid.numpy().reshape(-1))), "valid_shape": tuple(valid.shape), "u": u0.clone(),
           "rendered_bands": acc["rendered"], "rendered_fg_bands": acc["rendered_fg"], "rendered_bg_bands": acc["rendered_bg"],
           "ties_fg": ties(acc["ties_fg"]), "ties_bg": ties(acc["ties_bg"])}
    path = os.path.join(OUT_DIR, tag + ".pt")
    torch.save(out, path)
    print(tag, "->", path, os.path.getsize(path) // 1024, "KiB", "fg valid frac", float(valid.float().mean()), "indices per field", out["inds_fg_u8"].numel(),
          "composite mask mean", float(torch.cat([r["mask"] for r in acc["rendered"]], 1).mean()))


def gen_ops(ns):
    """Op-level goldens straight from the reference functions."""
    g = torch.Generator().manual_seed(7)
    out = {}
    # PosEmbedding incl. annealing window (embedding.py:69-125; reference test: tests/test_ops.py:64-133)
    x = torch.randn(5, 7, 3, generator=g) * 0.3
    for L, alpha in [(10, None), (12, 0.37), (6, 0.9), (0, None)]:
        pe = ns.embedding.PosEmbedding(3, L)
        pe.set_alpha(alpha)
        out[f"posenc_L{L}_a{alpha}"] = (x, pe(x).clone())
    # quaternion algebra (quat_transform.py)
    a, b = torch.randn(11, 4, generator=g), torch.randn(11, 4, generator=g)
    v = torch.randn(11, 3, generator=g)
    qt = ns.quat_transform
    out["qmul44"] = (a, b, qt.quaternion_mul(a, b))
    out["qmul43"] = (a, v, qt.quaternion_mul(a, v))
    out["qmul34"] = (v, b, qt.quaternion_mul(v, b))
    out["qconj"] = (a, qt.quaternion_conjugate(a))
    out["qapply"] = (a, v, qt.quaternion_apply(a, v))
    dq1, dq2 = (a, b), (torch.randn(11, 4, generator=g), torch.randn(11, 4, generator=g))
    out["dqmul"] = (dq1, dq2, qt.dual_quaternion_mul(dq1, dq2))
    out["dqapply"] = (dq1, v, qt.dual_quaternion_apply(dq1, v))
    out["dq2qt"] = (dq1, qt.dual_quaternion_to_quaternion_translation(dq1))
    # compositing (render_utils.py)
    dens = torch.rand(2, 5, 9, 1, generator=g) * 30
    deltas = torch.rand(2, 5, 9, 1, generator=g) * 0.05
    w, t = ns.render_utils.compute_weights(dens, deltas)
    out["compute_weights"] = (dens, deltas, w, t)
    bins = torch.sort(torch.rand(6, 15, generator=g), -1)[0]
    wts = torch.rand(6, 14, generator=g)
    wts[2] = 0  # degenerate ray: every bin has weight 0
    wts[3, 3:9] = 0
    captured = {}
    _ss = torch.searchsorted

    def searchsorted(cdf, u, right=False):
        r = _ss(cdf, u, right=right)
        captured["inds"] = r.clone()
        return r

    ns.render_utils.torch.searchsorted = searchsorted
    try:
        s = ns.render_utils.sample_pdf(bins, wts, 16, det=True)
    finally:
        ns.render_utils.torch.searchsorted = _ss
    out["sample_pdf"] = (bins, wts, s, captured["inds"])
    hxy = torch.cat([torch.rand(2, 4, 2, generator=g) * 64, torch.ones(2, 4, 1)], -1)
    Kinv = torch.linalg.inv(torch.tensor([[64.0, 0, 32], [0, 64, 32], [0, 0, 1]]))[None].repeat(2, 1, 1)
    nf = torch.tensor([[0.4, 0.8], [0.5, 0.9]])
    out["sample_cam_rays"] = (hxy, Kinv, nf, ns.render_utils.sample_cam_rays(hxy, Kinv, nf, n_depth=7))
    # compose_fields (multifields.py:339-398) for two fake fields
    fdA = {"density": torch.rand(2, 3, 4, 1, generator=g), "rgb": torch.rand(2, 3, 4, 3, generator=g),
           "depth": torch.sort(torch.rand(2, 3, 4, 1, generator=g), 2)[0], "cyc_dist": torch.rand(2, 3, 4, 1, generator=g)}
    fdB = {"density": torch.rand(2, 3, 4, 1, generator=g), "rgb": torch.rand(2, 3, 4, 3, generator=g),
           "depth": torch.sort(torch.rand(2, 3, 4, 1, generator=g), 2)[0]}
    dA, dB = torch.rand(2, 3, 4, 1, generator=g), torch.rand(2, 3, 4, 1, generator=g)
    comp, dcomp = ns.multifields.MultiFields.compose_fields({"fg": dict(fdA), "bg": dict(fdB)}, {"fg": dA, "bg": dB})
    out["compose_fields"] = (fdA, fdB, dA, dB, {k: v.clone() for k, v in comp.items()}, dcomp)
    path = os.path.join(OUT_DIR, "ops.pt")
    torch.save(out, path)
    print("ops ->", path, os.path.getsize(path) // 1024, "KiB")

def gen_comp_warp(ns):
    """ComposedWarp (skeleton skinning + dense post-warp, fg_motion "comp_skel-quad_dense", warping.py:143-170,445-483):
    backward warp, forward warp, and the frame_id=None forward warp that skips the post-warp."""
    P = synthetic.add_dense_weights(synthetic.make_weights(0))
    torch.manual_seed(0)
    di = ref_shim.synthetic_data_info(64)
    f = ns.deformable.Deformable("comp_skel-quad_dense", di, num_freq_dir=-1, appr_channels=32, num_inst=1, init_scale=0.2)
    f.category = "fg"
    sd = {k: v for k, v in P.items() if k in f.state_dict()}
    missing = [k for k in P if k not in f.state_dict() and k != "warp.skinning_model.symm_idx"]
    assert not missing, missing
    f.load_state_dict(sd, strict=False)
    M, N, D = 2, 4, 6
    fr = synthetic.make_frames(41, M, 64)
    fr["frame_id"] = torch.tensor([3, 4])
    fr = frames_from_reference(f, fr)
    with torch.no_grad():
        fr["t_embed_dense"] = f.warp.post_warp.time_embedding(fr["frame_id"]).clone()
    fr = synthetic.add_codes(fr, P)
    g = torch.Generator().manual_seed(43)
    xyz = (torch.randn(M, N, D, 3, generator=g) * 0.06).requires_grad_(True)
    w = torch.randn(M, N, D, 3, generator=g)
    sdict = {"t_articulation": fr["t_articulation"], "rest_articulation": fr["rest_articulation"]}
    out_bw, aux_bw = f.warp(xyz, fr["frame_id"], fr["inst_id"], backward=True, samples_dict=sdict, return_aux=True)
    out_fw, aux_fw = f.warp(xyz, fr["frame_id"], fr["inst_id"], backward=False, samples_dict=sdict, return_aux=True)
    out_fw_none = f.warp(xyz, None, fr["inst_id"], backward=False, samples_dict=sdict)
    dense_fw = f.warp.post_warp(xyz, fr["frame_id"], fr["inst_id"], backward=False)
    dense_bw = f.warp.post_warp(xyz, fr["frame_id"], fr["inst_id"], backward=True)
    loss = (out_bw * w).sum() + (out_fw * w.flip(0)).sum()
    names = ["warp.post_warp.forward_map.linear_1.0.weight", "warp.post_warp.backward_map.linear_2.0.weight",
             "warp.post_warp.backward_map.linear_final.bias", "warp.skinning_model.delta_field.linear_1.0.weight"]
    params = dict(f.named_parameters())
    grads = torch.autograd.grad(loss, [xyz] + [params[n] for n in names])
    out = {"weight_checksum": weight_checksum(P), "frames": {k: v for k, v in fr.items()}, "xyz": xyz.detach(), "w": w,
           "out_bw": out_bw.detach(), "out_fw": out_fw.detach(), "out_fw_none": out_fw_none.detach(),
           "dense_fw": dense_fw.detach(), "dense_bw": dense_bw.detach(),
           "aux_bw": {k: v.detach() for k, v in aux_bw.items()}, "loss": loss.detach(),
           "grad_xyz": grads[0], "grads": {n: compress_grad(gv) for n, gv in zip(names, grads[1:])}}
    path = os.path.join(OUT_DIR, "comp_warp.pt")
    torch.save(out, path)
    print("comp_warp ->", path, os.path.getsize(path) // 1024, "KiB")


def gen_bg_field(ns):
    """NeRF.forward of the background field NeRF(num_freq_xyz=6, num_freq_dir=0, appr_channels=0) (multifields.py:86-93):
    rgb / density / sdf for random points and view directions, gradients wrt inputs and a few weights."""
    P = synthetic.make_bg_weights(0)
    torch.manual_seed(0)
    di = ref_shim.synthetic_data_info(64)
    f = ns.nerf.NeRF(di, num_freq_xyz=6, num_freq_dir=0, appr_channels=0, init_scale=0.1)
    f.category = "bg"
    sd = {k: v for k, v in P.items() if k in f.state_dict()}
    missing = [k for k in P if k not in f.state_dict()]
    assert not missing, missing
    f.load_state_dict(sd, strict=False)
    M, N, D = 2, 5, 7
    g = torch.Generator().manual_seed(47)
    xyz = (torch.randn(M, N, D, 3, generator=g) * 0.3).requires_grad_(True)
    dirs = torch.randn(M, N, D, 3, generator=g)
    dirs = (dirs / dirs.norm(dim=-1, keepdim=True)).requires_grad_(True)
    frame_id, inst_id = torch.tensor([3, 4]), torch.zeros(2, dtype=torch.long)
    rgb, density = f(xyz, dir=dirs, frame_id=frame_id, inst_id=inst_id)
    sdf = f(xyz, inst_id=inst_id, get_density=False)
    w = torch.randn(M, N, D, 3, generator=g)
    w1 = torch.randn(M, N, D, 1, generator=g)
    loss = (rgb * w).sum() + (density * w1).sum() * 0.01
    names = ["basefield.linear_1.0.weight", "basefield.linear_5.0.weight", "colorfield.linear_1.0.weight", "rgb.0.weight", "rgb.2.bias",
             "sdf.weight", "colorfield.inst_embedding.mapping.weight"]
    params = dict(f.named_parameters())
    grads = torch.autograd.grad(loss, [xyz, dirs] + [params[n] for n in names])
    out = {"weight_checksum": weight_checksum(P), "xyz": xyz.detach(), "dir": dirs.detach(), "w": w, "w1": w1, "rgb": rgb.detach(),
           "density": density.detach(), "sdf": sdf.detach(), "loss": loss.detach(), "grad_xyz": grads[0], "grad_dir": grads[1],
           "grads": {n: compress_grad(gv) for n, gv in zip(names, grads[2:])}}
    path = os.path.join(OUT_DIR, "bg_field.pt")
    torch.save(out, path)
    print("bg_field ->", path, os.path.getsize(path) // 1024, "KiB")


def gen_comp_eval(ns):
    """field_type "comp" in eval mode, the way dvr_model.render_samples drives it (engine/model.py:328-361): fg
    Deformable("skel-quad").query_field + bg NeRF.query_field on the same rays -> MultiFields.compose_fields -> render_pixel of
    the composite and of each field."""
    M, N, D, res, seed = 2, 6, 16, 64, 61
    Pf = synthetic.make_weights(seed, sdf_bias=-0.02)
    f = build_reference_field(ns, Pf)
    f.eval()
    frf = frames_from_reference(f, synthetic.make_frames(seed + 1, M, res))
    Pb = synthetic.make_bg_weights(seed)
    Pb["sdf.bias"] = torch.tensor([-0.1])  # some occupancy in the bg so the composite has contributions from both fields
    torch.manual_seed(0)
    di = ref_shim.synthetic_data_info(64)
    b = ns.nerf.NeRF(di, num_freq_xyz=6, num_freq_dir=0, appr_channels=0, init_scale=0.1)
    b.category = "bg"
    missing = [k for k in Pb if k not in b.state_dict()]
    assert not missing, missing
    b.load_state_dict({k: v for k, v in Pb.items()}, strict=False)
    b.eval()
    frb = synthetic.make_bg_frames(seed + 3, M, res)
    g = torch.Generator().manual_seed(seed + 2)
    hxy = torch.cat([torch.rand(M, N, 2, generator=g) * res, torch.ones(M, N, 1)], -1)
    for fld in (f, b):
        orig = fld.importance_sampling
        fld.importance_sampling = (lambda o: (lambda *a, **k: o(*a, n_depth=D, **k)))(orig)
    sdf_ = samples_dict_of(frf, hxy, None)
    del sdf_["feature"]
    sdb = {"Kinv": frb["Kinv"], "field2cam": frb["field2cam"], "frame_id": frb["frame_id"], "inst_id": frb["inst_id"],
           "near_far": frb["near_far"], "hxy": hxy}
    fd_f, d_f, _ = f.query_field(sdf_)
    fd_b, d_b, _ = b.query_field(sdb)
    comp, dcomp = ns.multifields.MultiFields.compose_fields({"fg": dict(fd_f), "bg": dict(fd_b)}, {"fg": d_f, "bg": d_b})
    rendered = ns.render_utils.render_pixel(comp, dcomp)
    r_f = ns.render_utils.render_pixel(fd_f, d_f)
    r_b = ns.render_utils.render_pixel(fd_b, d_b)
    det = lambda d: {k: v.detach() for k, v in d.items()}
    out = {"meta": {"M": M, "N": N, "D": D, "res": res, "seed": seed, "bg_sdf_bias": -0.1, "fg_sdf_bias": -0.02,
                    "weight_checksum_fg": weight_checksum(Pf), "weight_checksum_bg": weight_checksum(Pb)},
           "frames_fg": {k: (tuple(t.detach() for t in v) if isinstance(v, tuple) else v.detach()) for k, v in frf.items()},
           "frames_bg": {k: (tuple(t.detach() for t in v) if isinstance(v, tuple) else v.detach()) for k, v in frb.items()},
           "hxy": hxy, "bg_feat_dict": det(fd_b), "bg_deltas": d_b.detach(), "rendered": det(rendered), "rendered_fg": det(r_f),
           "rendered_bg": det(r_b), "composed_depth": comp["depth"].detach(), "composed_keys": sorted(comp.keys())}
    path = os.path.join(OUT_DIR, "comp_eval.pt")
    torch.save(out, path)
    print("comp_eval ->", path, os.path.getsize(path) // 1024, "KiB", "keys", sorted(comp.keys()), "bg mask", float(r_b["mask"].mean()),
          "fg mask", float(r_f["mask"].mean()))


def gen_comp_train(ns, tag="comp_train", M=2, N=6, D=8, res=64, seed=71, fg_motion="skel-quad", frame_id=None, full_grid_stride=None, rows=None):
    """field_type "comp", training mode: fg Deformable.query_field + bg NeRF.query_field -> compose_fields -> render_pixel ->
    dvr_model.compute_recon_loss / mask_losses / apply_loss_weights with config field_type = "comp", plus gradients of the
    total loss wrt fg and bg weights.  Pins the oracle (render_train_comp / recon_losses_comp) and, through it and
    directly, the device path (tests/test_gpu_field.py).
    full_grid_stride / rows (round 4): the same graph at BASELINE configs[2]'s per-GPU shape -- a band of image rows of a res x res frame pair,
    D samples per ray and field, fg_motion "comp_skel-human_dense"; every stride-th ray of the three renders is stored, gradients of EVERY
    fg / bg weight (compressed), inputs are regenerated from the seeds by the tests."""
    num_bones = 18 if "skel-human" in fg_motion else 25
    Pf = synthetic.make_weights(seed, num_bones=num_bones)
    if fg_motion.startswith("comp_"):
        Pf = synthetic.add_dense_weights(Pf, seed, 1)
    f = build_reference_field(ns, Pf, 1, fg_motion)
    f.train()
    fr0 = synthetic.make_frames(seed + 1, M, res, num_bones=num_bones)
    if frame_id is not None:
        fr0["frame_id"] = torch.tensor(frame_id, dtype=torch.long)
    frf = frames_from_reference(f, fr0)
    Pb = synthetic.make_bg_weights(seed)
    Pb["sdf.bias"] = torch.tensor([-0.1])
    torch.manual_seed(0)
    di = ref_shim.synthetic_data_info(64)
    b = ns.nerf.NeRF(di, num_freq_xyz=6, num_freq_dir=0, appr_channels=0, init_scale=0.1)
    b.category = "bg"
    b.load_state_dict({k: v for k, v in Pb.items()}, strict=False)
    b.train()
    frb = synthetic.make_bg_frames(seed + 3, M, res)
    g = torch.Generator().manual_seed(seed + 2)
    if full_grid_stride:
        hxy = synthetic.make_rays(res, M, rows=rows)
        N = hxy.shape[1]
    else:
        hxy = torch.cat([torch.rand(M, N, 2, generator=g) * res, torch.ones(M, N, 1)], -1)
    batch = synthetic.make_targets(seed + 3, M, N, res, hxy)
    eik_inds = torch.randperm(M * N, generator=g)[: max(M * N // 16, 1)]
    match_perm = torch.randperm(M * N * D, generator=g)[: min(1024, M * N * D)]
    ns.nerf.torch.multinomial = lambda probs, n, replacement=False: eik_inds.clone()
    ns.feature.torch = ns.nerf.torch.__class__(**vars(ns.nerf.torch))
    ns.feature.torch.randperm = lambda n: torch.cat([match_perm, torch.arange(n)])
    ns.nerf.sample_cam_rays = partial(ns.render_utils.sample_cam_rays, n_depth=D)
    sdf_ = samples_dict_of(frf, hxy, batch["feature"])
    sdb = {"Kinv": frb["Kinv"], "field2cam": frb["field2cam"], "frame_id": frb["frame_id"], "inst_id": frb["inst_id"],
           "near_far": frb["near_far"], "hxy": hxy}
    fd_f, d_f, aux = f.query_field(sdf_, flow_thresh=float(res))
    fd_b, d_b, _ = b.query_field(sdb, flow_thresh=float(res))
    comp, dcomp = ns.multifields.MultiFields.compose_fields({"fg": dict(fd_f), "bg": dict(fd_b)}, {"fg": d_f, "bg": d_b})
    rendered = ns.render_utils.render_pixel(comp, dcomp)
    aux_fg = dict(aux)
    aux_fg.update(ns.render_utils.render_pixel(fd_f, d_f))
    aux_bg = ns.render_utils.render_pixel(fd_b, d_b)
    results = {"rendered": dict(rendered), "aux_dict": {"fg": aux_fg, "bg": dict(aux_bg)}}
    results["rendered"]["xyz_matches"] = aux["xyz_matches"]
    results["rendered"]["xyz_reproj"] = aux["xyz_reproj"]
    ref_out = {"rendered": {k: v.detach().clone() for k, v in results["rendered"].items()},
               "aux_fg": {k: v.detach().clone() for k, v in aux_fg.items()}, "aux_bg": {k: v.detach().clone() for k, v in aux_bg.items()}}
    import importlib
    model = importlib.import_module("lab4d.engine.model").dvr_model
    config = {"field_type": "comp", "train_res": res}
    loss_dict = {}
    model.compute_recon_loss(loss_dict, results, batch, config)  # (scales aux_dict["bg"]["vis"] by 0.01 in place)
    model.mask_losses(loss_dict, batch, config)
    if "gauss_mask" not in rendered:
        assert fg_motion in ("rigid", "dense")  # deformable.py:344: the gaussian-bone density exists for SkinningWarp only
    loss_dict["reg_eikonal"] = rendered["eikonal"]
    loss_dict["reg_deform_cyc"] = aux_fg["cyc_dist"]
    loss_dict["reg_delta_skin"] = aux_fg["delta_skin"]
    loss_dict["reg_skin_entropy"] = aux_fg["skin_entropy"]
    from oracle.lab4d_oracle import DEFAULT_LOSS_WT
    config.update(DEFAULT_LOSS_WT)
    model.apply_loss_weights(loss_dict, config)
    total = sum(loss_dict.values())
    pf, pb = dict(f.named_parameters()), dict(b.named_parameters())
    if full_grid_stride:  # every weight of the path
        fnames = [k for k in Pf if k in pf and pf[k].requires_grad]
        bnames = [k for k in Pb if k in pb and pb[k].requires_grad]
    else:
        fnames = ["basefield.linear_1.0.weight", "rgb.0.weight", "warp.skinning_model.delta_field.linear_1.0.weight", "sdf.weight"]
        bnames = ["basefield.linear_1.0.weight", "basefield.linear_5.0.weight", "colorfield.linear_2.0.weight", "rgb.0.weight", "sdf.weight",
                  "vis_mlp.basefield.linear_1.0.weight"]
    grads = torch.autograd.grad(total, [pf[k] for k in fnames] + [pb[k] for k in bnames], allow_unused=True)
    gd = {}
    for k, gv in zip(["fg:" + k for k in fnames] + ["bg:" + k for k in bnames], grads):
        if gv is not None:
            gd[k] = compress_grad(gv.detach())
    out = {"meta": {"M": M, "N": N, "D": D, "res": res, "seed": seed, "bg_sdf_bias": -0.1, "flow_thresh": float(res), "fg_motion": fg_motion,
                    "weight_checksum_fg": weight_checksum(Pf), "weight_checksum_bg": weight_checksum(Pb)},
           "frames_fg": {k: (tuple(t.detach() for t in v) if isinstance(v, tuple) else v.detach()) for k, v in frf.items()},
           "frames_bg": {k: (tuple(t.detach() for t in v) if isinstance(v, tuple) else v.detach()) for k, v in frb.items()},
           "hxy": hxy, "batch": batch, "rng": {"eik_inds": eik_inds.clone(), "eik_inds_bg": eik_inds.clone(), "match_perm": match_perm.clone()},
           "bg_feat_dict": {k: v.detach() for k, v in fd_b.items()}, **ref_out,
           "loss": {k: v.detach() for k, v in loss_dict.items()}, "grads": gd}
    if full_grid_stride:
        st = full_grid_stride
        out["meta"]["full_grid_stride"] = st
        out["meta"]["rows"] = rows
        for k in ("hxy", "batch", "bg_feat_dict"):
            out.pop(k)
        for name in ("rendered", "aux_fg", "aux_bg"):
            out[name] = {k: (v[:, ::st].clone() if v.dim() >= 2 and v.shape[1] == N else v) for k, v in out[name].items()}
    path = os.path.join(OUT_DIR, tag + ".pt")
    torch.save(out, path)
    print(tag, "->", path, os.path.getsize(path) // 1024, "KiB", {k: round(float(v), 6) for k, v in loss_dict.items()})


def main(only=None):
    """Regenerate every fixture (or the named subset: tests/test_golden_generator.py keeps this recipe from rotting)."""
    ns = ref_shim.load()
    # give render_utils a private torch namespace so searchsorted can be observed
    import types
    ns.render_utils.torch = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
    jobs = [
        ("ops", lambda: gen_ops(ns)),
        ("train_small", lambda: gen_train(ns, "small", M=2, N=6, D=8, res=64, seed=11)),
        ("train_alpha", lambda: gen_train(ns, "alpha", M=4, N=5, D=6, res=64, seed=21, alpha=0.45)),
        ("train_compmotion", lambda: gen_train(ns, "compmotion", M=2, N=6, D=8, res=64, seed=51, fg_motion="comp_skel-quad_dense", frame_id=[3, 4])),
        # BASELINE configs[2]'s fg field: the 18-joint human skeleton with the dense post-warp (fg_motion "comp_skel-human_dense")
        ("train_human", lambda: gen_train(ns, "human", M=2, N=6, D=8, res=64, seed=81, fg_motion="comp_skel-human_dense", frame_id=[10, 11])),
        # fg_motion "rigid" (the reference's default, config.py:42; IdentityWarp) and "dense" (a bare DenseWarp, D=6)
        ("train_rigid", lambda: gen_train(ns, "rigid", M=2, N=6, D=8, res=64, seed=91, fg_motion="rigid")),
        ("train_dense", lambda: gen_train(ns, "dense", M=2, N=6, D=8, res=64, seed=101, fg_motion="dense", frame_id=[7, 8])),
        ("train_multi", lambda: gen_train(ns, "multi", M=4, N=5, D=6, res=64, seed=31, num_inst=3, inst_id=[1, 1, 2, 2], frame_id=[22, 23, 44, 45])),
        # BASELINE configs[3]'s field: 10 videos (num_inst=10, per-instance codes in every CondMLP) with fg_motion comp_skel-quad_dense; a pair of video 3
        # and a pair of video 7 (frames 21-22 / 46-47 of the 64-frame, 10-video synthetic set)
        ("train_multi10", lambda: gen_train(ns, "multi10", M=4, N=5, D=6, res=64, seed=111, num_inst=10, inst_id=[3, 3, 7, 7], frame_id=[21, 22, 46, 47],
                                            fg_motion="comp_skel-quad_dense")),
        # BASELINE config 0: 64x64 crop x 64 samples
        ("train_c1", lambda: gen_train(ns, "c1", M=2, N=None, D=64, res=64, seed=41, full_grid_stride=16)),
        # BASELINE config 1 = the bench shape: 512x512, 128 samples/ray; a 2-row band (rows 255-256: half inside the target mask's
        # disc) of a frame pair = 2 x 1,024 rays = 262,144 samples through the reference; every 16th ray is stored
        ("train_bench", lambda: gen_train(ns, "bench", M=2, N=None, D=128, res=512, seed=61, full_grid_stride=16, rows=(255, 257))),
        ("eval_small", lambda: gen_eval(ns, "small", M=2, N=8, D=16, res=64, seed=31)),
        ("eval_rigid", lambda: gen_eval(ns, "rigid", M=2, N=8, D=16, res=64, seed=33, fg_motion="rigid")),
        ("eval_dense", lambda: gen_eval(ns, "dense", M=2, N=8, D=16, res=64, seed=35, fg_motion="dense")),
        ("comp_warp", lambda: gen_comp_warp(ns)),
        ("bg_field", lambda: gen_bg_field(ns)),
        ("comp_eval", lambda: gen_comp_eval(ns)),
        ("comp_train", lambda: gen_comp_train(ns)),
        # round 4: BASELINE configs[2] / configs[3] at their per-GPU shapes (like train_bench for configs[1]): a 2-row band of a 512x512 frame pair.
        # comp_bench: MultiFields "comp" with fg_motion comp_skel-human_dense (18 bones + dense post-warp) + bg, 64 + 64 samples per ray;
        # train_multi10_bench: the 10-instance category field (comp_skel-quad_dense), 128 samples per ray, a pair of video 3
        ("comp_bench", lambda: gen_comp_train(ns, "comp_bench", M=2, N=None, D=64, res=512, seed=131, fg_motion="comp_skel-human_dense", frame_id=[10, 11],
                                              full_grid_stride=16, rows=(255, 257))),
        ("train_multi10_bench", lambda: gen_train(ns, "multi10_bench", M=2, N=None, D=128, res=512, seed=121, num_inst=10, inst_id=[3, 3], frame_id=[21, 22],
                                                  fg_motion="comp_skel-quad_dense", full_grid_stride=16, rows=(255, 257))),
        # round 5.  w1_weights: SURVEY 8d's fitted weight set (the reference's own geometry_init on seed 61's W0).  train_bench_w1: the bench-shape
        # training fixture on it.  eval_bench / eval_bench_w1: the eval path at the bench size -- 8 rows of a 512x512 pair = 8,192 rays, 524,288
        # importance indices, 1,048,576 valid-mask bits -- on W0 (+ sdf bias nudge, like eval_small) and on W1.
        ("w1_weights", lambda: gen_w1_weights(ns)),
        ("train_bench_w1", lambda: gen_train(ns, "bench_w1", M=2, N=None, D=128, res=512, seed=61, full_grid_stride=16, rows=(255, 257), w1=True)),
        ("eval_bench", lambda: gen_eval_bench(ns, "bench")),
        ("eval_bench_w1", lambda: gen_eval_bench(ns, "bench_w1", w1=True)),
        # ... and the comp configuration's eval path at configs[2]'s shape: fg comp_skel-human_dense + bg, 32 + 32 samples per field, 4,096 rays
        ("comp_eval_bench", lambda: gen_comp_eval_bench(ns)),
    ]
    for name, job in jobs:
        if only is None or name in only:
            job()


if __name__ == "__main__":
    main(sys.argv[1:] or None)
