"""End-to-end GPU parity of the training graph (rays -> warp -> MLPs -> compositing -> losses -> gradients)
against golden vectors produced by the REFERENCE's own code (tests/golden/train_*.pt) and the oracle."""
import os

import pytest
import torch

from lab4d_amd import synthetic
from oracle import lab4d_oracle as O
from parity_report import FLOOR_FACTOR_FULL, check, report

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


# Bound per rendered channel, as a fraction of the channel's largest reference value, fp32 path.  Everything is held to 3e-4
# except the eikonal channel: (|d sdf/dx| - 1)^2 is a squared first derivative through 9 layers and a 2^f-weighted sum over
# the positional-encoding slots, and the REFERENCE'S OWN fp32 arithmetic is 6.8e-4 of the channel maximum away from the same
# graph evaluated in fp64 (tests/golden/fp32_noise_floor.json, written by tests/measure_fp32_noise_floor.py): no fp32
# implementation with a different accumulation order can agree with it more closely than that.
RENDER_TOL_F32 = {"eikonal": 2e-3}
# importance-sampling indices that may differ from the reference's (by one bin, at a cdf entry equal to the query to rounding)
# on eval_small.pt: 16 rays x 8 fine samples = 128 indices.  Measured on MI355X: see gpurun_out/parity_eval_indices.json.
EVAL_INDEX_MISMATCH_MAX = 2  # measured: 1 of 128




def load_case(golden_dir, name):
    from fixture_utils import fg_weights
    g = torch.load(os.path.join(golden_dir, name), weights_only=False)
    return g, fg_weights(g["meta"])


@pytest.mark.parametrize("case", ["train_small.pt", "train_alpha.pt", "train_compmotion.pt", "train_human.pt", "train_rigid.pt", "train_dense.pt",
                                  "train_multi10.pt"])
def test_training_graph_matches_reference_goldens(golden_dir, case):
    """train_multi10: BASELINE configs[3]'s field -- 10 instances (per-instance codes in every CondMLP), fg_motion comp_skel-quad_dense, pairs of two videos.
    train_rigid / train_dense: fg_motion "rigid" (the reference's default, IdentityWarp) and "dense" (a bare 6-layer DenseWarp, LAB4D_NET_DENSE6);
    their identically-zero terms are NaN losses in the reference (the mean of an empty selection) and here."""
    from lab4d_amd import deformable as DF
    g, P = load_case(golden_dir, case)
    meta = g["meta"]
    Pd = {k: (v.to(DEV).clone().requires_grad_(True) if v.dtype.is_floating_point and k != "aabb" else v.to(DEV)) for k, v in P.items()}
    fr = synthetic.to_device(dict(g["frames"]), DEV)
    leaves = {}
    for n in ["Kinv", "field2cam", "t_articulation", "rest_articulation"]:
        v = fr[n]
        if isinstance(v, tuple):
            v = tuple(t.clone().requires_grad_(True) for t in v)
            for i, t in enumerate(v):
                leaves[f"{n}.{i}"] = t
        else:
            v = v.clone().requires_grad_(True)
            leaves[n] = v
        fr[n] = v
    fr = synthetic.add_codes(fr, Pd)
    batch = synthetic.to_device(g["batch"], DEV)
    fr["feature"] = batch["feature"]
    rng = synthetic.to_device(g["rng"], DEV)
    hxy = g["hxy"].to(DEV)
    fd, deltas, aux = DF.query_field_train(Pd, fr, hxy, rng, flow_thresh=meta["flow_thresh"], n_depth=meta["D"], alpha=meta["alpha"])
    measured = {}
    for k, v in g["feat_dict"].items():
        measured["feat_dict." + k] = rel(fd[k], v)
    measured["deltas"] = rel(deltas, g["deltas"])
    res = DF.render_train(Pd, fr, hxy, rng, flow_thresh=meta["flow_thresh"], n_depth=meta["D"], alpha=meta["alpha"])
    for k, v in g["rendered"].items():
        measured["rendered." + k] = rel(res["rendered"][k], v)
    for k, v in g["aux_fg"].items():
        measured["aux_fg." + k] = rel(res["aux_dict"]["fg"][k], v)
    # PSNR of the rendered colour against the reference render (north_star: "matched PSNR")
    mse = float(((res["rendered"]["rgb"].detach().cpu() - g["rendered"]["rgb"]) ** 2).mean())
    assert mse < 1e-9, f"rgb PSNR {(-10 * torch.log10(torch.tensor(mse))).item():.1f} dB"
    losses = DF.losses_fg(res, batch, meta["res"], DF.DEFAULT_LOSS_WT)
    assert set(losses) == set(g["loss"]), (sorted(losses), sorted(g["loss"]))
    for k, v in g["loss"].items():
        if bool(torch.isnan(v)):
            assert bool(torch.isnan(losses[k])), k
        else:
            measured["loss." + k] = rel(losses[k], v)
    total = sum(v for v in losses.values() if bool(torch.isfinite(v)))
    names = [k for k in g["grads"] if not k.startswith("frame:")]
    fnames = [k[6:] for k in g["grads"] if k.startswith("frame:")]
    grads = torch.autograd.grad(total, [Pd[k] for k in names] + [leaves[k] for k in fnames], allow_unused=True)
    for k, gv in zip(names + ["frame:" + k for k in fnames], grads):
        ref = g["grads"][k]
        assert gv is not None, k
        if k.startswith("frame:rest_articulation"):
            # The two forward warps of a training query share one evaluation of the skinning field (deformable.query_field_train,
            # "rest_shared_in_pair"): the rest articulation is the same tensor for both frames of a pair (the fixtures', like
            # any reference batch's, is identical row for row), so what is defined is the gradient summed over the pair --
            # the reference attributes the flow branch's share to the partner's copy of the row, this path to the own copy.
            pair = lambda t: t.reshape(t.shape[0] // 2, 2, -1).sum(1)
            e = rel(pair(gv), pair(ref["full"]))
        elif "full" in ref:
            e = rel(gv, ref["full"])
        else:
            e = rel(gv.flatten()[:: ref["stride"]], ref["sub"])
        measured["gradmax." + k] = e
    # bound per entry = max(1e-4, 2 x the committed MI355X measurement); what exceeds 1e-4 must lie within 4x of the reference's own fp32
    # noise floor on this fixture (tests/parity_report.py).  Per-frame input gradients ("frame:") have no floor entry: they are held to
    # the measurement alone.
    frame = {k: v for k, v in measured.items() if k.startswith("gradmax.frame:")}
    check("small_" + case[:-3], {k: v for k, v in measured.items() if k not in frame}, floor_case=case[:-3])
    check("small_frames_" + case[:-3], frame)


def test_unshared_forward_warps_give_the_reference_rows(golden_dir):
    """With fr["rest_shared_in_pair"] = False both forward warps evaluate their own skinning field, exactly like the reference:
    the per-ROW gradient of rest_articulation then matches the fixture as well (the shared path only matches the pair sums)."""
    from lab4d_amd import deformable as DF
    g, P = load_case(golden_dir, "train_small.pt")
    meta = g["meta"]
    Pd = synthetic.to_device(P, DEV)
    fr = synthetic.to_device(dict(g["frames"]), DEV)
    rest = tuple(t.clone().requires_grad_(True) for t in fr["rest_articulation"])
    fr["rest_articulation"] = rest
    fr = synthetic.add_codes(fr, Pd)
    batch = synthetic.to_device(g["batch"], DEV)
    fr["feature"] = batch["feature"]
    fr["rest_shared_in_pair"] = False
    res = DF.render_train(Pd, fr, g["hxy"].to(DEV), synthetic.to_device(g["rng"], DEV), flow_thresh=meta["flow_thresh"], n_depth=meta["D"], alpha=meta["alpha"])
    measured = {"rendered." + k: rel(res["rendered"][k], v) for k, v in g["rendered"].items()}
    grads = torch.autograd.grad(DF.losses_fg(res, batch, meta["res"], DF.DEFAULT_LOSS_WT).total, list(rest))
    frame = {"gradmax.frame:rest_articulation.%d" % i: rel(gv, g["grads"]["frame:rest_articulation.%d" % i]["full"]) for i, gv in enumerate(grads)}
    check("unshared_train_small", measured, floor_case="train_small")
    check("unshared_frames_train_small", frame)  # per-frame input gradients have no floor entry: held to the committed measurement


def test_training_graph_multi_instance(golden_dir):
    """BASELINE config 4's shape: 3 instances, two frame pairs from different videos, per-instance codes in every CondMLP.  The
    oracle is pinned to this reference-generated fixture on the CPU (tests/test_oracle_golden.py)."""
    test_training_graph_matches_reference_goldens(golden_dir, "train_multi.pt")


def _run_full_size(golden_dir, name, prec):
    """A reference-generated fixture at a BASELINE size (rays / targets regenerated from the seeds, every stride-th ray of the
    reference's render stored): device render + losses + gradients, returned as measured errors relative to the reference.
    rendered.* / loss.*: max abs error over the largest reference value; grad.*: relative L2 error of the (sub-sampled) tensor."""
    from lab4d_amd import deformable as DF
    from fixture_utils import fg_weights
    g = torch.load(os.path.join(golden_dir, name), weights_only=False)
    meta = g["meta"]
    st, M, res, seed = meta["full_grid_stride"], meta["M"], meta["res"], meta["seed"]
    P = fg_weights(meta)
    Pd = {k: (v.to(DEV).clone().requires_grad_(True) if v.dtype.is_floating_point and k != "aabb" else v.to(DEV)) for k, v in P.items()}
    hxy = synthetic.make_rays(res, M, rows=meta.get("rows"))
    batch = synthetic.to_device(synthetic.make_targets(seed + 3, M, hxy.shape[1], res, hxy), DEV)
    fr = synthetic.add_codes(synthetic.to_device(dict(g["frames"]), DEV), Pd)
    fr["feature"] = batch["feature"]
    out = DF.render_train(Pd, fr, hxy.to(DEV), synthetic.to_device(g["rng"], DEV), flow_thresh=meta["flow_thresh"], n_depth=meta["D"],
                          alpha=meta["alpha"], prec=prec)
    measured = {}
    for k, v in g["rendered"].items():
        measured["rendered." + k] = rel(out["rendered"][k][:, ::st], v)
    mse = float(((out["rendered"]["rgb"][:, ::st].detach().cpu() - g["rendered"]["rgb"]) ** 2).mean())
    measured["psnr_rgb_db"] = -10.0 * torch.log10(torch.tensor(max(mse, 1e-20))).item()
    losses = DF.losses_fg(out, batch, res, DF.DEFAULT_LOSS_WT)
    for k, v in g["loss"].items():
        measured["loss." + k] = rel(losses[k], v)
    names = [k for k in g["grads"] if not k.startswith("frame:")]
    grads = torch.autograd.grad(sum(losses.values()), [Pd[k] for k in names], allow_unused=True)
    for k, gv in zip(names, grads):
        ref = g["grads"][k]
        assert gv is not None, k
        a, b = (gv, ref["full"]) if "full" in ref else (gv.flatten()[:: ref["stride"]], ref["sub"])
        a, b = a.detach().double().cpu().flatten(), b.double().flatten()
        measured["grad." + k] = float((a - b).norm() / (b.norm() + 1e-30))
    return measured


def _assert_bounds(tag, measured, bounds, default):
    report(tag, measured)
    bad = {}
    for k, e in measured.items():
        if k == "psnr_rgb_db":
            continue
        family = k.split(".")[0]
        b = bounds.get(k, bounds.get(family, default))
        if not e < b:
            bad[k] = (e, b)
    assert not bad, bad


# fp32 path: every entry (rendered channel, loss, gradient tensor in relative L2) is held to max(1e-4, 2 x the committed MI355X measurement), and
# whatever exceeds 1e-4 -- the colour net's gradients (3e-4 .. 8e-4), the eikonal channel -- to within 4x of the REFERENCE'S OWN float32-vs-float64
# deviation on the same fixture (tests/golden/fp32_noise_floor.json: colourfield gradients 4e-4 .. 8e-4, eikonal 2e-3 at this size): the
# reference's arithmetic itself is that far from exact there, no fp32 implementation with another accumulation order can agree more closely.


def test_training_graph_at_baseline_config0_size(golden_dir):
    """BASELINE.json configs[0] at full size on the device: the 64x64 crop of a frame pair x 64 samples/ray (524,288 samples)
    against the reference-generated fixture (every 16th ray of the render, losses, compressed gradients); fp32 path."""
    from lab4d_amd import mlp
    check("config0_fp32", _run_full_size(golden_dir, "train_c1.pt", mlp.PREC_F32), floor_case="train_c1", skip=("psnr_rgb_db",), floor_factor=FLOOR_FACTOR_FULL)


def test_training_graph_at_the_bench_shape_fp32(golden_dir):
    """BASELINE.json configs[1]'s shape (the shape bench.py times): 512x512 frame pair, 128 samples/ray -- a 2-row band of both
    frames (2,048 rays, 262,144 samples) through the whole training graph against the reference's own output; fp32 path."""
    from lab4d_amd import mlp
    check("bench_fp32", _run_full_size(golden_dir, "train_bench.pt", mlp.PREC_F32), floor_case="train_bench", skip=("psnr_rgb_db",), floor_factor=FLOOR_FACTOR_FULL)


def test_training_graph_at_the_bench_shape_w1_fp32(golden_dir):
    """Round 5: the bench-shape training fixture on W1 -- seed 61's weights after the reference's own geometry_init (nerf.py:251-295 with the Gaussian-bone
    SDF of deformable.py:95-117; tests/golden/make_golden.py: gen_w1_weights): a fitted, sharp surface, where compositing weights are peaked."""
    from lab4d_amd import mlp
    check("bench_w1_fp32", _run_full_size(golden_dir, "train_bench_w1.pt", mlp.PREC_F32), floor_case="train_bench_w1", skip=("psnr_rgb_db",), floor_factor=FLOOR_FACTOR_FULL)


def test_training_graph_at_the_bench_shape_w1_bf16(golden_dir):
    from lab4d_amd import mlp
    m = _run_full_size(golden_dir, "train_bench_w1.pt", mlp.PREC_BF16)
    check("bench_w1_bf16", m, skip=("psnr_rgb_db",))
    assert m["psnr_rgb_db"] > 60.0, m["psnr_rgb_db"]


def test_training_graph_at_the_bench_shape_multi10_fp32(golden_dir):
    """BASELINE.json configs[3]'s field at the bench shape (round 4): 10 instances, fg_motion comp_skel-quad_dense, a 2-row band of a 512x512 pair of
    video 3 x 128 samples/ray against the reference's own output; every entry within max(1e-4, 2 x measured) AND, above 1e-4, within 4x of the
    SAME tensor's fp32-vs-fp64 floor on this fixture (no pooled family floor at full size)."""
    from lab4d_amd import mlp
    check("multi10_bench_fp32", _run_full_size(golden_dir, "train_multi10_bench.pt", mlp.PREC_F32), floor_case="train_multi10_bench", skip=("psnr_rgb_db",),
          floor_factor=FLOOR_FACTOR_FULL)


def _run_comp(golden_dir, name, prec):
    """field_type "comp", training mode on the device against a reference-generated fixture (comp_train.pt: 12 rays; comp_bench.pt: BASELINE
    configs[2]'s per-GPU shape, every 16th ray stored): the three renders, the comp losses, gradients of the fg and bg weights.
    rendered.* / aux_*.* / loss.*: max abs error over the largest reference value; grad.*: relative L2 (of the sub-sampled tensor where the
    fixture stores one); gradmax.*: max abs error over the largest reference entry (fully stored tensors)."""
    from lab4d_amd import deformable as DF
    from fixture_utils import bg_weights, fg_weights, leaf, rays_and_targets, strided
    g = torch.load(os.path.join(golden_dir, name), weights_only=False)
    meta = g["meta"]
    Pf, Pb = leaf(fg_weights(meta), DEV), {k: v.to(DEV).clone().requires_grad_(True) for k, v in bg_weights(meta).items()}
    hxy, batch = rays_and_targets(g)
    batch = synthetic.to_device(batch, DEV)
    frf = synthetic.add_codes(synthetic.to_device(dict(g["frames_fg"]), DEV), Pf)
    frf["feature"] = batch["feature"]
    frb = synthetic.add_bg_codes(synthetic.to_device(dict(g["frames_bg"]), DEV), Pb)
    rng = synthetic.to_device(g["rng"], DEV)
    hxy = hxy.to(DEV)
    measured = {}
    if "bg_feat_dict" in g:
        fd_b, _, _ = DF.query_field_train_bg(Pb, frb, hxy, rng, flow_thresh=meta["flow_thresh"], n_depth=meta["D"], prec=prec)
        assert sorted(fd_b.keys()) == sorted(g["bg_feat_dict"].keys())
        for k, v in g["bg_feat_dict"].items():
            measured["bg_feat_dict." + k] = rel(fd_b[k], v)
    res = DF.render_train_comp(Pf, frf, Pb, frb, hxy, rng, flow_thresh=meta["flow_thresh"], n_depth=meta["D"], prec=prec)
    for name_, ref in (("rendered", g["rendered"]), ("aux_fg", g["aux_fg"]), ("aux_bg", g["aux_bg"])):
        got = res["rendered"] if name_ == "rendered" else res["aux_dict"][name_[4:]]
        for k, v in ref.items():
            measured[name_ + "." + k] = rel(strided(g, got[k]), v)
    mse = float(((strided(g, res["rendered"]["rgb"]).detach().cpu() - g["rendered"]["rgb"]) ** 2).mean())
    measured["psnr_rgb_db"] = -10.0 * torch.log10(torch.tensor(max(mse, 1e-20))).item()
    losses = DF.losses_comp(res, batch, meta["res"], DF.DEFAULT_LOSS_WT)
    for k, v in g["loss"].items():
        measured["loss." + k] = rel(losses[k], v)
    names = list(g["grads"].keys())
    grads = torch.autograd.grad(sum(losses.values()), [(Pf if n.startswith("fg:") else Pb)[n[3:]] for n in names], allow_unused=True)
    for n, gv in zip(names, grads):
        ref = g["grads"][n]
        assert gv is not None, n
        a, b = (gv, ref["full"]) if "full" in ref else (gv.flatten()[:: ref["stride"]], ref["sub"])
        a, b = a.detach().double().cpu().flatten(), b.double().flatten()
        measured["grad." + n] = float((a - b).norm() / (b.norm() + 1e-30))
        if "full" in ref:
            measured["gradmax." + n] = float((a - b).abs().max() / (b.abs().max() + 1e-30))
    return measured


def test_comp_training_graph_at_the_bench_shape_fp32(golden_dir):
    """BASELINE.json configs[2] at its per-GPU shape (round 4): MultiFields "comp" with fg_motion comp_skel-human_dense + bg, a 2-row band of a
    512x512 frame pair, 64 + 64 samples per ray composed, against the reference's own renders / losses / gradients of EVERY fg and bg weight
    (tests/golden/comp_bench.pt); same-tensor floors (tests/golden/fp32_noise_floor.json: comp_bench)."""
    from lab4d_amd import mlp
    check("comp_bench_fp32", _run_comp(golden_dir, "comp_bench.pt", mlp.PREC_F32), floor_case="comp_bench", skip=("psnr_rgb_db",), floor_factor=FLOOR_FACTOR_FULL)


def test_comp_training_graph_at_the_bench_shape_bf16(golden_dir):
    """The benched dtype on the same fixture: bounds = BF16_BOUNDS (measured margins, see below), PSNR of the composite colour vs the reference."""
    from lab4d_amd import mlp
    m = _run_comp(golden_dir, "comp_bench.pt", mlp.PREC_BF16)
    _assert_bounds("comp_bench_bf16_ceiling", {k: v for k, v in m.items() if not k.startswith("gradmax.")}, BF16_BOUNDS_COMP, 2e-2)
    check("comp_bench_bf16", {k: v for k, v in m.items() if not k.startswith("gradmax.")}, skip=("psnr_rgb_db",))  # per entry: 2 x the committed measurement
    assert m["psnr_rgb_db"] > 80.0, m["psnr_rgb_db"]


# bf16 path (the dtype bench.py times; BASELINE configs[1] says bf16).  MFMA operands -- weights and every stored activation --
# are rounded to 8 significant bits (unit roundoff u = 2^-9 = 2.0e-3), accumulation is fp32.  A rendered channel passes through
# up to 10 (sdf) + 5 (colour) such layers; rounding errors are independent per layer, so the expected relative error of a
# per-sample output is ~ sqrt(15) u = 8e-3; compositing 128 samples (weights sum to <= 1, errors independent per sample)
# averages that down by ~ sqrt(128), to the 1e-4 level the geometry / colour channels show.  Bounds = the values measured on
# MI355X for this fixture (profiles/r02_parity_bench_bf16.json; worst rendered channel 7.6e-5, worst gradient 1.1e-2) with a
# 2-3x margin.  Channels with their own rows: eikonal ((|d sdf/dx| - 1)^2: a squared derivative, 8.3e-3 measured), the
# 16-channel feature field and the soft-argmax match built on it (L2-normalised 128-wide MLP output, 3.3e-3 / 1.7e-3), and
# delta_skin (mean square of a 64-wide MLP's output, 8.3e-4).  Gradients: relative L2 per parameter tensor (the dgrad chain
# sees bf16 weights and bf16 stored dZ; wgrad contracts bf16 dZ with bf16 activations in fp32).
BF16_BOUNDS = {
    "rendered": 2e-4, "rendered.eikonal": 2e-2, "rendered.feature": 1e-2, "rendered.xyz_matches": 5e-3, "rendered.delta_skin": 3e-3,
    "loss": 1.5e-4, "loss.reg_eikonal": 2e-3, "loss.reg_delta_skin": 1e-3, "grad": 2.5e-2,
}


# the comp configuration's rows: the bg scene spans |xyz| ~ 0.6 (its visibility / eikonal channels see the 2^9 posenc band), the composite mixes two fields
# measured on MI355X (profiles/r04_parity_comp_bench_bf16.json): composite / per-field colour, depth, geometry <= 2.2e-4; the skinning-derived channels of the
# 18-bone field (cyc_dist, skin_entropy, delta_skin, gauss_mask) 1e-3 .. 2e-3; eikonal 4e-3 .. 9e-3; gradients <= 2.9e-2 relative L2 (basefield.linear_4)
BF16_BOUNDS_COMP = dict(BF16_BOUNDS, **{"rendered": 5e-4, "rendered.mask_bg": 2e-3, "rendered.cyc_dist": 3e-3, "rendered.skin_entropy": 5e-3, "rendered.delta_skin": 5e-3,
                                      "rendered.gauss_mask": 4e-3, "rendered.eikonal": 2e-2, "rendered.feature": 1e-2, "rendered.xyz_matches": 5e-3,
                                      "aux_fg": 5e-4, "aux_fg.cyc_dist": 3e-3, "aux_fg.skin_entropy": 5e-3, "aux_fg.delta_skin": 5e-3, "aux_fg.gauss_mask": 4e-3,
                                      "aux_fg.eikonal": 2e-2, "aux_fg.feature": 1e-2, "aux_fg.xyz_matches": 5e-3, "aux_bg": 5e-4, "aux_bg.eikonal": 3e-2,
                                      "loss": 3e-4, "loss.reg_eikonal": 2e-3, "loss.reg_delta_skin": 4e-3, "loss.reg_deform_cyc": 2e-3, "grad": 6e-2})


def test_training_graph_at_the_bench_shape_bf16(golden_dir):
    """The benched dtype at the benched shape against the reference (fp32) render of the same rays / weights."""
    from lab4d_amd import mlp
    m = _run_full_size(golden_dir, "train_bench.pt", mlp.PREC_BF16)
    _assert_bounds("bench_bf16_ceiling", m, BF16_BOUNDS, 2e-2)   # the a-priori ceilings of the analysis below ...
    check("bench_bf16", m, skip=("psnr_rgb_db",))                # ... and, per entry, max(1e-4, 2 x the committed MI355X measurement) (round 5)
    assert m["psnr_rgb_db"] > 90.0, m["psnr_rgb_db"]  # measured 99.3 dB vs the reference render


def test_bf16_training_graph_is_close_to_fp32(golden_dir):
    """bf16 MFMA path: rendered colour within PSNR > 35 dB of the fp32 reference render, mask within 2e-2."""
    from lab4d_amd import deformable as DF
    from lab4d_amd import mlp
    g, P = load_case(golden_dir, "train_small.pt")
    meta = g["meta"]
    Pd = synthetic.to_device(P, DEV)
    fr = synthetic.add_codes(synthetic.to_device(dict(g["frames"]), DEV), Pd)
    batch = synthetic.to_device(g["batch"], DEV)
    fr["feature"] = batch["feature"]
    res = DF.render_train(Pd, fr, g["hxy"].to(DEV), synthetic.to_device(g["rng"], DEV), flow_thresh=meta["flow_thresh"], n_depth=meta["D"],
                          alpha=meta["alpha"], prec=mlp.PREC_BF16)
    mse = float(((res["rendered"]["rgb"].detach().cpu() - g["rendered"]["rgb"]) ** 2).mean())
    psnr = -10 * torch.log10(torch.tensor(mse)).item()
    assert psnr > 35, psnr
    assert rel(res["rendered"]["mask"], g["rendered"]["mask"]) < 3e-2


def test_eval_graph_matches_reference_goldens(golden_dir):
    """Eval mode: importance sampling (indices), valid mask, normals, rendered channels vs the reference's own output."""
    from lab4d_amd import deformable as DF
    g, P = load_case(golden_dir, "eval_small.pt")
    meta = g["meta"]
    Pd = synthetic.to_device(P, DEV)
    fr = synthetic.add_codes(synthetic.to_device(dict(g["frames"]), DEV), Pd)
    out = DF.render_eval(Pd, fr, g["hxy"].to(DEV), n_depth=meta["D"])
    inds = out["debug"]["inds"].cpu()
    mism = (inds != g["inds"]).float().mean().item()
    # The density that feeds sample_pdf went through the warp and 10 fp32 MFMA layers, so a cdf entry can land on the other
    # side of a query u where the two agree to rounding: such an index differs by exactly one bin and the interpolated sample
    # is the same point (the rendered channels below are the functional check).  The measured rate on this fixture is
    # printed / stored next to the bound; the bound is the fixture's measured count + 1 sample, not a blanket percentage.
    n_diff = int((inds != g["inds"]).sum())
    valid = out["debug"]["valid"].cpu()
    vm = int((valid != g["valid"]).sum())
    report("eval_indices", {"importance_index_mismatch_rate": mism, "importance_index_mismatch_count": float(n_diff),
                            "importance_index_total": float(inds.numel()), "valid_mask_mismatch_count": float(vm)})
    assert n_diff <= EVAL_INDEX_MISMATCH_MAX, (n_diff, inds.numel())
    assert int((inds - g["inds"]).abs().max()) <= 1
    assert vm == 0, "valid mask must be identical (bit-exact bool) on this fixture"
    # index differences are confined to exact cdf ties (the u = 1 end point): the resulting samples coincide, so every
    # rendered channel must still match the reference render at 1e-4 (fp32 path)
    check("eval_small", {"rendered." + k: rel(out["rendered"][k], v) for k, v in g["rendered"].items()}, floor_case="eval_small")


def _run_eval_bench(golden_dir, name, prec):
    """An eval_bench fixture on the device, band by band like the reference rendered it: returns (measured render errors over the stored rays of all
    bands, index statistics, mask mismatches)."""
    from lab4d_amd import deformable as DF
    from fixture_utils import cdf_of_weights, check_index_mismatches, eval_bench_bands, eval_bench_unpack
    g, P = load_case(golden_dir, name)
    meta = g["meta"]
    M, D, st = meta["M"], meta["D"], meta["full_grid_stride"]
    Pd = synthetic.to_device(P, DEV)
    fr = synthetic.add_codes(synthetic.to_device(dict(g["frames"]), DEV), Pd)
    inds_ref, valid_ref = eval_bench_unpack(g)
    got, ref = {}, {}
    stats = {"index_total": 0, "index_mismatch": 0, "index_cdf_gap_max": 0.0, "valid_total": 0, "valid_mismatch": 0, "valid_fraction": 0.0}
    for band, hxy, sl in eval_bench_bands(g):
        out = DF.render_eval(Pd, fr, hxy.to(DEV), n_depth=D, prec=prec)
        n = hxy.shape[1]
        inds = out["debug"]["inds"].cpu().view(M, n, D // 2)
        valid = out["debug"]["valid"].cpu().view(M, n, D)
        stats["index_total"] += inds.numel()
        stats["valid_total"] += valid.numel()
        stats["valid_mismatch"] += int((valid != valid_ref[:, sl]).sum())
        stats["valid_fraction"] += float(valid.float().sum())
        if prec == 0:  # fp32: every differing index must be a one-bin shift at a near tie of the reference's cdf, the two cdfs equal to the fp32 floor
            cdf = cdf_of_weights(out["debug"]["weights_coarse"].reshape(M * n, D // 2)).view(M, n, -1)
            k, gap = check_index_mismatches(g, band, inds, cdf, tol=INDEX_CDF_TOL)
        else:
            k, gap = int((inds != inds_ref[:, sl]).sum()), 0.0
        stats["index_mismatch"] += k
        stats["index_cdf_gap_max"] = max(stats["index_cdf_gap_max"], gap)
        for ch, v in g["rendered_bands"][band].items():
            got.setdefault(ch, []).append(out["rendered"][ch][:, ::st].detach().float().cpu())
            ref.setdefault(ch, []).append(v)
    stats["valid_fraction"] /= stats["valid_total"]
    measured = {"rendered." + ch: rel(torch.cat(got[ch], 1), torch.cat(ref[ch], 1)) for ch in got}
    # the rendered normal is normalize(sum_d w_d n_d / (sum_d w_d + 1e-6)) (render_utils.py:59-96): on a ray that hits nothing it is the direction of a sum of
    # rounding errors -- any two evaluations point anywhere (the reference's own fp32 vs fp64: 4e-2 of a UNIT vector as a max over 512 rays, a heavy-tailed
    # statistic).  It is compared where it means something: weighted by the ray's opacity (the reference's rendered mask), in this test and in the floor
    # (tests/measure_fp32_noise_floor.py: eval_bench_case) alike
    # ... and as a relative L2 over the stored rays, not a maximum: on the raw-initialisation fixture (W0: ten octaves of positional encoding through
    # random weights) the gradient's direction is a heavy-tailed function of the sample position, and the maximum over 512 rays of the device-vs-reference
    # difference (7.5e-2, call 3) sat 2.95x above the maximum of the fp32-vs-fp64 difference -- two draws of the same tail; W1's fitted surface: 5.6e-3
    m_ref = torch.cat(ref["mask"], 1).double()
    nd, nr = torch.cat(got["normal"], 1).double() * m_ref, torch.cat(ref["normal"], 1).double() * m_ref
    measured["rendered.normal"] = float((nd - nr).norm() / (nr.norm() + 1e-30))
    mse = float(((torch.cat(got["rgb"], 1) - torch.cat(ref["rgb"], 1)) ** 2).mean())
    measured["psnr_rgb_db"] = -10.0 * torch.log10(torch.tensor(max(mse, 1e-20))).item()
    return measured, stats


# |cdf_device - cdf_reference| allowed at a cdf entry where the two implementations' importance indices differ: the cdf is a running sum of <= 62
# normalised weights, each carrying the fp32 rounding of ten 256-wide layers upstream.  Measured worst on MI355X (profiles/r05_parity_eval_bench*_indices_fp32.json):
# 2.4e-7 (W0) / 7.7e-7 (W1) -- one to six ulps of a number near 1
INDEX_CDF_TOL = 2e-6
# importance indices that may differ from the reference's -- EACH ONE asserted to be a one-bin shift at a near tie whose two cdfs agree to INDEX_CDF_TOL
# (fixture_utils.check_index_mismatches) -- as a fraction of the number the REFERENCE'S OWN arithmetic disagrees with itself on when evaluated in float32
# and in float64 on the same fixture (tests/golden/fp32_noise_floor.json: index_mismatch_count*): an exact tie in one fp32 evaluation order is broken the
# other way by any other order.  Measured on MI355X: eval_bench 1,387 of 524,288 (floor 2,915: 0.48), eval_bench_w1 1,546 (2,806: 0.55),
# comp_eval_bench 758 + 709 of 262,144 (1,325 + 1,457: 0.53)
INDEX_MISMATCH_VS_FLOOR_MAX = 0.75


def _index_floor(case):
    import json
    fl = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fp32_noise_floor.json")))[case]
    return sum(v for k, v in fl.items() if k.startswith("index_mismatch_count"))


@pytest.mark.parametrize("name", ["eval_bench", "eval_bench_w1"])
def test_eval_graph_at_the_bench_size_fp32(golden_dir, name):
    """Round 5: the eval path -- importance sampling, compute_normal on every sample, get_valid_idx + query_nerf compaction -- at the
    size bench.py's eval leg runs (512x512, 64 + 64 samples per ray), 8 image rows of a frame pair = 8,192 rays, against the reference's own output:
    * the valid mask, 1,048,576 bits: BIT-EXACT;
    * the importance indices, 524,288: every index that differs from the reference's is ASSERTED to be a one-bin shift at a cdf entry the fixture lists
      as a near tie (within 1e-4 of a query point u), where the device's cdf (re-formed from the device's coarse weights with the arithmetic the
      kernel is held to bit for bit) and the reference's straddle u and agree to INDEX_CDF_TOL; their number is bounded and reported;
    * every rendered channel over the stored rays: max(1e-4, 2 x measured) AND within 2x of the same channel's fp32-vs-fp64 floor on this fixture.
    eval_bench: raw seeded init + sdf bias nudge (a flat field); eval_bench_w1: W1, the reference's geometry_init fit (a sharp surface)."""
    from lab4d_amd import mlp
    measured, stats = _run_eval_bench(golden_dir, name + ".pt", mlp.PREC_F32)
    report(name + "_indices_fp32", {k: float(v) for k, v in stats.items()})
    assert stats["valid_total"] >= 1_000_000 and stats["index_total"] >= 500_000
    assert stats["valid_mismatch"] == 0, "valid mask must be identical (bit-exact bool): %d of %d differ" % (stats["valid_mismatch"], stats["valid_total"])
    assert stats["index_mismatch"] <= INDEX_MISMATCH_VS_FLOOR_MAX * _index_floor(name), (stats, _index_floor(name))
    check(name + "_fp32", measured, floor_case=name, skip=("psnr_rgb_db",), floor_factor=FLOOR_FACTOR_FULL)


# bf16 eval path (what bench.py's eval leg times) against the fp32 reference render: the coarse densities carry bf16 rounding, so importance samples
# move by more than a tie (indices are reported, not asserted) and the valid mask -- a function of the warped sample POSITIONS, which bf16 touches only
# through the delta-skin MLP -- may flip for samples within rounding of a box face.  Bounds = 2 x the MI355X measurement (profiles/r05_parity_eval_bench*_bf16.json).


@pytest.mark.parametrize("name", ["eval_bench", "eval_bench_w1"])
def test_eval_graph_at_the_bench_size_bf16(golden_dir, name):
    from lab4d_amd import mlp
    measured, stats = _run_eval_bench(golden_dir, name + ".pt", mlp.PREC_BF16)
    report(name + "_indices_bf16", {k: float(v) for k, v in stats.items()})
    check(name + "_bf16", measured, skip=("psnr_rgb_db",))
    assert stats["valid_mismatch"] <= 1e-4 * stats["valid_total"], stats
    assert measured["psnr_rgb_db"] > 40.0, measured["psnr_rgb_db"]


def _run_comp_eval_bench(golden_dir, prec):
    """comp_eval_bench.pt on the device, band by band: (measured render errors of the composite / fg / bg renders over the stored rays, index and
    mask statistics of both fields)."""
    from lab4d_amd import deformable as DF
    from fixture_utils import bg_weights, cdf_of_weights, check_index_mismatches, eval_bench_bands, fg_weights, unpack_bits
    g = torch.load(os.path.join(golden_dir, "comp_eval_bench.pt"), weights_only=False)
    meta = g["meta"]
    M, D, st = meta["M"], meta["D"], meta["full_grid_stride"]
    Pf, Pb = synthetic.to_device(fg_weights(meta), DEV), synthetic.to_device(bg_weights(meta), DEV)
    frf = synthetic.add_codes(synthetic.to_device(dict(g["frames_fg"]), DEV), Pf)
    frb = synthetic.add_bg_codes(synthetic.to_device(dict(g["frames_bg"]), DEV), Pb)
    valid_ref = unpack_bits(g["valid_bits"], g["valid_shape"])
    refs = {"fg": g["inds_fg_u8"].long(), "bg": g["inds_bg_u8"].long()}
    got, ref = {}, {}
    stats = {"index_total": 0, "index_mismatch_fg": 0, "index_mismatch_bg": 0, "index_cdf_gap_max": 0.0, "valid_total": 0, "valid_mismatch": 0}
    for band, hxy, sl in eval_bench_bands(g):
        out = DF.render_eval_comp(Pf, frf, Pb, frb, hxy.to(DEV), n_depth=D, prec=prec)
        n = hxy.shape[1]
        valid = out["debug"]["fg"]["valid"].cpu().view(M, n, D)
        stats["valid_total"] += valid.numel()
        stats["valid_mismatch"] += int((valid != valid_ref[:, sl]).sum())
        for fld in ("fg", "bg"):
            inds = out["debug"][fld]["inds"].cpu().view(M, n, D // 2)
            stats["index_total"] += inds.numel()
            if prec == 0:
                cdf = cdf_of_weights(out["debug"][fld]["weights_coarse"].reshape(M * n, D // 2)).view(M, n, -1)
                k, gap = check_index_mismatches(g, band, inds, cdf, tol=INDEX_CDF_TOL, ref=refs[fld], ties=g["ties_" + fld])
            else:
                k, gap = int((inds != refs[fld][:, sl]).sum()), 0.0
            stats["index_mismatch_" + fld] += k
            stats["index_cdf_gap_max"] = max(stats["index_cdf_gap_max"], gap)
        for name, bands in (("rendered", g["rendered_bands"]), ("fg", g["rendered_fg_bands"]), ("bg", g["rendered_bg_bands"])):
            r = out["rendered"] if name == "rendered" else out["aux_dict"][name]
            for ch, v in bands[band].items():
                got.setdefault((name, ch), []).append(r[ch][:, ::st].detach().float().cpu())
                ref.setdefault((name, ch), []).append(v)
    measured = {}
    for (name, ch) in got:
        a, b = torch.cat(got[(name, ch)], 1), torch.cat(ref[(name, ch)], 1)
        if ch == "normal":  # opacity-weighted relative L2 (see _run_eval_bench)
            m_ref = torch.cat(ref[(name, "mask")], 1).double()
            a, b = a.double() * m_ref, b.double() * m_ref
            measured["%s.%s" % (name, ch)] = float((a - b).norm() / (b.norm() + 1e-30))
        else:
            measured["%s.%s" % (name, ch)] = rel(a, b)
    return measured, stats


def test_comp_eval_graph_at_the_bench_size_fp32(golden_dir):
    """Round 5: the comp configuration's eval path at BASELINE configs[2]'s per-GPU shape (fg comp_skel-human_dense + bg, 32 + 32 samples per field,
    compose_fields, three renders; 4,096 rays) against the reference: fg valid mask bit-exact (262,144 bits); both fields' importance indices
    (2 x 131,072) with every difference asserted to be a one-bin shift at a near tie (see test_eval_graph_at_the_bench_size_fp32); every
    channel of the three renders held to max(1e-4, 2 x measured) and to the same quantity's fp32-vs-fp64 floor."""
    from lab4d_amd import mlp
    measured, stats = _run_comp_eval_bench(golden_dir, mlp.PREC_F32)
    report("comp_eval_bench_indices_fp32", {k: float(v) for k, v in stats.items()})
    assert stats["valid_mismatch"] == 0, stats
    assert stats["index_mismatch_fg"] + stats["index_mismatch_bg"] <= INDEX_MISMATCH_VS_FLOOR_MAX * _index_floor("comp_eval_bench"), stats
    check("comp_eval_bench_fp32", measured, floor_case="comp_eval_bench", floor_factor=FLOOR_FACTOR_FULL)


def test_comp_eval_graph_at_the_bench_size_bf16(golden_dir):
    from lab4d_amd import mlp
    measured, stats = _run_comp_eval_bench(golden_dir, mlp.PREC_BF16)
    report("comp_eval_bench_indices_bf16", {k: float(v) for k, v in stats.items()})
    check("comp_eval_bench_bf16", measured)
    # bf16 moves the warped sample positions through the delta-skin and dense post-warp nets: samples within rounding of a box face flip (measured: 32 of 262,144)
    assert stats["valid_mismatch"] <= 2.5e-4 * stats["valid_total"], stats


def test_render_samples_chunk_equals_unchunked_eval(golden_dir):
    """Chunking along N is invisible in eval mode (model.py:259-326), apart from the global mean(T) normaliser of `vis`."""
    from lab4d_amd import model
    g, P = load_case(golden_dir, "eval_small.pt")
    Pd = synthetic.to_device(P, DEV)
    sd = synthetic.to_device(dict(g["frames"]), DEV)
    sd["hxy"] = g["hxy"].to(DEV)
    full = model.render_samples(Pd, sd, training=False, n_depth=g["meta"]["D"])
    chunked = model.render_samples_chunk(Pd, sd, chunk_size=6, training=False, n_depth=g["meta"]["D"])
    for k, v in full["rendered"].items():
        if k == "vis":
            continue
        assert chunked["rendered"][k].shape == v.shape
        assert rel(chunked["rendered"][k].cpu(), v.cpu()) < 1e-5, k


def test_comp_eval_matches_reference_goldens(golden_dir):
    """field_type "comp" in eval mode on the device: bg NeRF.query_field, compose_fields, render_pixel of the composite and
    of both fields against the reference-generated fixture (fp32)."""
    from lab4d_amd import deformable as DF
    g = torch.load(os.path.join(golden_dir, "comp_eval.pt"), weights_only=False)
    meta = g["meta"]
    Pf = synthetic.to_device(synthetic.make_weights(meta["seed"], sdf_bias=meta["fg_sdf_bias"]), DEV)
    Pb = synthetic.make_bg_weights(meta["seed"])
    Pb["sdf.bias"] = torch.tensor([meta["bg_sdf_bias"]])
    Pb = synthetic.to_device(Pb, DEV)
    frf = synthetic.add_codes(synthetic.to_device(dict(g["frames_fg"]), DEV), Pf)
    frb = synthetic.add_bg_codes(synthetic.to_device(dict(g["frames_bg"]), DEV), Pb)
    hxy = g["hxy"].to(DEV)
    fd_b, d_b, _ = DF.query_field_eval_bg(Pb, frb, hxy, n_depth=meta["D"])
    assert sorted(fd_b.keys()) == sorted(g["bg_feat_dict"].keys())
    # the bg scene spans |xyz| ~ 0.6: the 2^9 posenc band of the visibility net amplifies the 1e-7 rounding differences of the
    # rigid transform (reference: torch-CPU quaternion ops) to ~1e-3 of its logits, and (|g|-1)^2 is ill-conditioned near
    # |g| = 1; colour / density / normals / geometry are held to 5e-4
    loose = ("vis", "eikonal", "normal")  # normal = normalised input gradient: gradient-class tolerance (cf. test_gpu_mlp: 2e-3)
    for k, v in g["bg_feat_dict"].items():
        assert rel(fd_b[k], v) < (5e-3 if k in loose else 5e-4), f"bg.{k}: {rel(fd_b[k], v):.3e}"
    res = DF.render_eval_comp(Pf, frf, Pb, frb, hxy, n_depth=meta["D"])
    assert sorted(res["composed"].keys()) == g["composed_keys"]
    assert rel(res["composed"]["depth"], g["composed_depth"]) < 2e-4  # importance samples move with the fp32 rounding of the coarse densities
    for name, ref in (("rendered", g["rendered"]), ("fg", g["rendered_fg"]), ("bg", g["rendered_bg"])):
        got = res["rendered"] if name == "rendered" else res["aux_dict"][name]
        for k, v in ref.items():
            assert rel(got[k], v) < (5e-3 if k in loose else 5e-4), f"{name}.{k}: {rel(got[k], v):.3e}"


def test_comp_train_matches_reference_goldens(golden_dir):
    """field_type "comp", training mode on the device (12 rays): bg NeRF.query_field (flow, eikonal through the tangent kernel of the bg
    basefield), compose_fields, the three renders, the comp losses and gradients wrt fg and bg weights vs the reference-generated fixture
    (fp32) -- round 4: in the regime of every other parity test (bound = max(1e-4, 2 x the committed MI355X measurement); what exceeds 1e-4 is
    held against the fp32-vs-fp64 floor of the same quantity, or of its family over the pool of 12-ray fixtures)."""
    from lab4d_amd import mlp
    m = _run_comp(golden_dir, "comp_train.pt", mlp.PREC_F32)
    check("comp_train_fp32", m, floor_case="comp_train", skip=("psnr_rgb_db",))


def test_render_samples_dispatches_comp(golden_dir):
    """model.render_samples with per-category samples_dict / parameters = dvr_model.render_samples for field_type "comp"."""
    from lab4d_amd import deformable as DF, model
    g = torch.load(os.path.join(golden_dir, "comp_eval.pt"), weights_only=False)
    meta = g["meta"]
    Pf = synthetic.to_device(synthetic.make_weights(meta["seed"], sdf_bias=meta["fg_sdf_bias"]), DEV)
    Pb = synthetic.make_bg_weights(meta["seed"])
    Pb["sdf.bias"] = torch.tensor([meta["bg_sdf_bias"]])
    Pb = synthetic.to_device(Pb, DEV)
    sf = synthetic.to_device(dict(g["frames_fg"]), DEV)
    sb = synthetic.to_device(dict(g["frames_bg"]), DEV)
    sf["hxy"] = sb["hxy"] = g["hxy"].to(DEV)
    res = model.render_samples({"fg": Pf, "bg": Pb}, {"fg": sf, "bg": sb}, training=False, n_depth=meta["D"])
    for k, v in g["rendered"].items():
        assert rel(res["rendered"][k], v) < (5e-3 if k in ("vis", "eikonal", "normal") else 5e-4), k
    assert set(res["aux_dict"].keys()) == {"fg", "bg"}
    chunked = model.render_samples_chunk({"fg": Pf, "bg": Pb}, {"fg": sf, "bg": sb}, chunk_size=6, training=False, n_depth=meta["D"])
    for k, v in res["rendered"].items():
        if k != "vis":  # vis is normalised by the mean transmittance of the (chunk of) rays
            assert rel(chunked["rendered"][k].cpu(), v.cpu()) < 1e-5, k


def _device_relu_pattern(masks, S, widths):
    """Decode the fp32 chain kernel's stored ReLU sign words (mlp_kernels.hpp: word [(tile * MT + mt) * 64 + lane], fp32 tiles of 32 samples;
    lane = (n, h), bit r <-> feature 32 mt + (r & 3) + 8 (r >> 2) + 4 h of sample 32 tile + n; 1 = alive) into per-layer (S, width) bools."""
    out = []
    r = torch.arange(16)
    for words, W in zip(masks, widths):
        if words is None:
            out.append(None)
            continue
        MT = W // 32
        w = words.cpu().view(-1, MT, 2, 32)  # (tile, mt, h, n)
        bits = ((w[..., None] >> r) & 1).bool()  # (tile, mt, h, n, r)
        feat = torch.zeros(MT, 2, 16, dtype=torch.long)
        for mt in range(MT):
            for h in range(2):
                feat[mt, h] = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h
        alive = torch.zeros(w.shape[0] * 32, W, dtype=torch.bool)
        samples = (torch.arange(w.shape[0])[:, None] * 32 + torch.arange(32)[None]).reshape(-1)  # (tile, n)
        for mt in range(MT):
            for h in range(2):
                alive[samples[:, None], feat[mt, h][None, :]] = bits[:, mt, h].reshape(-1, 16)
        out.append(alive[:S])
    return out


@pytest.mark.parametrize("case", ["train_small.pt", "train_alpha.pt", "train_multi10.pt", "train_human.pt", "train_compmotion.pt"])
def test_relu_patterns_of_the_small_fixtures_agree_with_the_oracle(golden_dir, case):
    """Rounds 2-3 explained the small fixtures' gradient deviations above 1e-4 by discrete events -- "a ReLU unit of the 8 x 256 basefield that lands on
    the other side of zero in two fp32 evaluations moves a gradient tensor by 1e-3 .. 1e-2" -- without testing it.  This is the test: on the
    reference's own canonical sample points (fixture feat_dict.xyz) the device's stored sign words (training-mode fp32 chain) and the oracle's
    pre-activations may differ in at most 2 of the ~220,000 hidden units, and only where the pre-activation is within 1e-5 of zero.  Measured:
    0 flips on all five fixtures (profiles/r04_parity_relu_flips_*.json) -- the story was wrong; the excesses came from the gradient subsample
    of the fixtures (tests/golden/make_golden.py: compress_grad) and are gone with it (tests/parity_report.py)."""
    import torch.nn.functional as F
    from lab4d_amd import mlp
    g, P = load_case(golden_dir, case)
    meta = g["meta"]
    xyz = g["feat_dict"]["xyz"]
    M, N, D = xyz.shape[:3]
    Pd = synthetic.to_device(P, DEV)
    fr = synthetic.add_codes(synthetic.to_device(dict(g["frames"]), DEV), Pd)
    x = xyz.reshape(-1, 3).to(DEV).clone().requires_grad_(True)
    tap = {}
    fw = None
    if meta.get("alpha") is not None:
        from lab4d_amd import deformable as DF
        fw = DF.posenc_window(meta["alpha"], 10, DEV)
    mlp.run_chain(mlp.NET_FG_BASE, mlp.PREC_F32, Pd, x, N * D, conds={0: fr["code_base"], 4: fr["code_base"]}, freq_w=fw, tap=tap)
    dev_alive = _device_relu_pattern(tap["masks"][:9], M * N * D, [256] * 9)
    # the oracle's pre-activations, layer by layer (base.py:65-78,123-150: input = [posenc | code], skip concatenates the input first)
    code = fr["code_base"].cpu()[:, None, None, :].expand(M, N, D, -1).reshape(M * N * D, -1)
    inp = torch.cat([O.pos_embedding(xyz.reshape(-1, 3), 10, meta.get("alpha")), code], -1)
    h, flips, near_zero, total = inp, 0, True, 0
    for i in range(9):
        name = "basefield.linear_%d.0" % (i + 1) if i < 8 else "basefield.linear_final.0"
        if i == 4:
            h = torch.cat([inp, h], -1)
        z = F.linear(h, P[name + ".weight"], P[name + ".bias"])
        h = F.relu(z)
        if dev_alive[i] is None:
            continue
        diff = dev_alive[i] != (z > 0)
        flips += int(diff.sum())
        total += diff.numel()
        if diff.any():
            near_zero = near_zero and bool((z[diff].abs() < 1e-5 * max(float(z.abs().max()), 1.0)).all())
    report("relu_flips_" + case[:-3], {"flipped_units": float(flips), "hidden_units": float(total)})
    assert total >= 8 * 256 * M * N * D and flips <= 2 and near_zero, (flips, total, near_zero)
