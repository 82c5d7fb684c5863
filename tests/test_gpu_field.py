"""End-to-end GPU parity of the training graph (rays -> warp -> MLPs -> compositing -> losses -> gradients)
against golden vectors produced by the REFERENCE's own code (tests/golden/train_*.pt) and the oracle."""
import os

import pytest
import torch

from lab4d_amd import synthetic
from oracle import lab4d_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


# Bound per rendered channel, as a fraction of the channel's largest reference value, fp32 path.  Everything is held to 3e-4
# except the eikonal channel: (|d sdf/dx| - 1)^2 is a squared first derivative through 9 layers and a 2^f-weighted sum over
# the positional-encoding slots, and the REFERENCE'S OWN fp32 arithmetic is 6.8e-4 of the channel maximum away from the same
# graph evaluated in fp64 (tests/test_oracle_properties.py::test_eikonal_fp32_noise_floor measures it): no fp32
# implementation with a different accumulation order can agree with it more closely than that.
RENDER_TOL_F32 = {"eikonal": 2e-3}


def report(tag, measured):
    """Measured errors go to stdout (pytest -s / -rP) and to gpurun_out/parity_<tag>.json so that the bound and the measurement
    can be read side by side."""
    import json
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_%s.json" % tag), "w") as f:
        json.dump({k: float("%.3e" % v) for k, v in measured.items()}, f, indent=1, sort_keys=True)
    print(tag, {k: "%.2e" % v for k, v in measured.items()})


def load_case(golden_dir, name):
    g = torch.load(os.path.join(golden_dir, name), weights_only=False)
    P = synthetic.make_weights(g["meta"]["seed"], num_inst=g["meta"].get("num_inst", 1), sdf_bias=g["meta"].get("sdf_bias"))
    return g, P


@pytest.mark.parametrize("case", ["train_small.pt", "train_alpha.pt"])
def test_training_graph_matches_reference_goldens(golden_dir, case):
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
    for k, v in g["feat_dict"].items():
        assert rel(fd[k], v) < 2e-4, f"feat_dict.{k}: {rel(fd[k], v):.3e}"
    assert rel(deltas, g["deltas"]) < 1e-5
    res = DF.render_train(Pd, fr, hxy, rng, flow_thresh=meta["flow_thresh"], n_depth=meta["D"], alpha=meta["alpha"])
    for k, v in g["rendered"].items():
        assert rel(res["rendered"][k], v) < 2e-4, f"rendered.{k}: {rel(res['rendered'][k], v):.3e}"
    for k, v in g["aux_fg"].items():
        assert rel(res["aux_dict"]["fg"][k], v) < 2e-4, f"aux_fg.{k}: {rel(res['aux_dict']['fg'][k], v):.3e}"
    # PSNR of the rendered colour against the reference render (north_star: "matched PSNR")
    mse = float(((res["rendered"]["rgb"].cpu() - g["rendered"]["rgb"]) ** 2).mean())
    assert mse < 1e-9, f"rgb PSNR {(-10 * torch.log10(torch.tensor(mse))).item():.1f} dB"
    losses = DF.losses_fg(res, batch, meta["res"], DF.DEFAULT_LOSS_WT)
    for k, v in g["loss"].items():
        assert rel(losses[k], v) < 5e-4, f"loss.{k}: {rel(losses[k], v):.3e}"
    total = sum(losses.values())
    names = [k for k in g["grads"] if not k.startswith("frame:")]
    fnames = [k[6:] for k in g["grads"] if k.startswith("frame:")]
    grads = torch.autograd.grad(total, [Pd[k] for k in names] + [leaves[k] for k in fnames], allow_unused=True)
    worst = 0.0
    for k, gv in zip(names + ["frame:" + k for k in fnames], grads):
        ref = g["grads"][k]
        assert gv is not None, k
        if "full" in ref:
            e = rel(gv, ref["full"])
        else:
            e = rel(gv.flatten()[:: ref["stride"]], ref["sub"])
        worst = max(worst, e)
        assert e < 5e-3, f"grad {k}: {e:.3e}"


def test_training_graph_multi_instance(golden_dir):
    """BASELINE config 4's shape: 3 instances, two frame pairs from different videos, per-instance codes in every CondMLP.  The
    oracle is pinned to this reference-generated fixture on the CPU (tests/test_oracle_golden.py)."""
    test_training_graph_matches_reference_goldens(golden_dir, "train_multi.pt")


def test_training_graph_at_baseline_config0_size(golden_dir):
    """BASELINE.json configs[0] at full size on the device: the 64x64 crop of a frame pair x 64 samples/ray (524,288 samples) against
    the reference-generated fixture (every 16th ray of the render, losses, compressed gradients); fp32 path."""
    from lab4d_amd import deformable as DF
    g = torch.load(os.path.join(golden_dir, "train_c1.pt"), weights_only=False)
    meta = g["meta"]
    st, M, res, seed = meta["full_grid_stride"], meta["M"], meta["res"], meta["seed"]
    P = synthetic.make_weights(seed)
    Pd = {k: (v.to(DEV).clone().requires_grad_(True) if v.dtype.is_floating_point and k != "aabb" else v.to(DEV)) for k, v in P.items()}
    hxy = synthetic.make_rays(res, M)
    batch = synthetic.to_device(synthetic.make_targets(seed + 3, M, hxy.shape[1], res, hxy), DEV)
    fr = synthetic.add_codes(synthetic.to_device(dict(g["frames"]), DEV), Pd)
    fr["feature"] = batch["feature"]
    out = DF.render_train(Pd, fr, hxy.to(DEV), synthetic.to_device(g["rng"], DEV), flow_thresh=meta["flow_thresh"], n_depth=meta["D"],
                          alpha=meta["alpha"])
    measured = {}
    for k, v in g["rendered"].items():
        e = measured["rendered." + k] = rel(out["rendered"][k][:, ::st], v)
        assert e < RENDER_TOL_F32.get(k, 3e-4), f"rendered.{k}: {e:.3e}"
    losses = DF.losses_fg(out, batch, res, DF.DEFAULT_LOSS_WT)
    for k, v in g["loss"].items():
        e = measured["loss." + k] = rel(losses[k], v)
        assert e < (2e-3 if k == "reg_eikonal" else 5e-4), f"loss.{k}: {e:.3e}"
    names = [k for k in g["grads"] if not k.startswith("frame:")]
    grads = torch.autograd.grad(sum(losses.values()), [Pd[k] for k in names], allow_unused=True)
    for k, gv in zip(names, grads):
        ref = g["grads"][k]
        assert gv is not None, k
        e = measured["grad." + k] = rel(gv, ref["full"]) if "full" in ref else rel(gv.flatten()[:: ref["stride"]], ref["sub"])
        assert e < 1e-2, f"grad {k}: {e:.3e}"
    report("config0_fp32", measured)


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
    mse = float(((res["rendered"]["rgb"].cpu() - g["rendered"]["rgb"]) ** 2).mean())
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
    # the density that feeds sample_pdf went through the warp and 10 fp32 MFMA layers: indices may differ from the
    # reference only where u falls within rounding of a cdf entry, i.e. by one bin and rarely; the
    # rendered channels below (2e-4) are the functional check
    assert mism <= 0.03, mism
    assert int((inds - g["inds"]).abs().max()) <= 1
    valid = out["debug"]["valid"].cpu()
    vm = (valid != g["valid"]).float().mean().item()
    assert vm <= 0.002, vm
    assert vm == 0.0, "valid mask must be identical on this fixture"
    # index differences are confined to exact cdf ties (the u = 1 end point): the resulting samples coincide, so every
    # rendered channel must still match the reference render at 1e-4 (fp32 path)
    for k, v in g["rendered"].items():
        assert rel(out["rendered"][k], v) < 2e-4, f"rendered.{k}: {rel(out['rendered'][k], v):.3e}"


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
    """field_type "comp", training mode on the device: bg NeRF.query_field (flow, eikonal through the tangent kernel of the bg
    basefield), compose_fields, the three renders, the comp losses and gradients wrt fg and bg weights vs the
    reference-generated fixture (fp32)."""
    from lab4d_amd import deformable as DF
    g = torch.load(os.path.join(golden_dir, "comp_train.pt"), weights_only=False)
    meta = g["meta"]
    Pf = synthetic.make_weights(meta["seed"])
    Pb = synthetic.make_bg_weights(meta["seed"])
    Pb["sdf.bias"] = torch.tensor([meta["bg_sdf_bias"]])
    Pf = {k: (v.to(DEV).clone().requires_grad_(True) if v.dtype.is_floating_point and k != "aabb" else v.to(DEV)) for k, v in Pf.items()}
    Pb = {k: v.to(DEV).clone().requires_grad_(True) for k, v in Pb.items()}
    frf = synthetic.add_codes(synthetic.to_device(dict(g["frames_fg"]), DEV), Pf)
    batch = synthetic.to_device(g["batch"], DEV)
    frf["feature"] = batch["feature"]
    frb = synthetic.add_bg_codes(synthetic.to_device(dict(g["frames_bg"]), DEV), Pb)
    rng = synthetic.to_device(g["rng"], DEV)
    hxy = g["hxy"].to(DEV)
    loose = ("vis", "eikonal")
    fd_b, _, _ = DF.query_field_train_bg(Pb, frb, hxy, rng, flow_thresh=meta["flow_thresh"], n_depth=meta["D"])
    assert sorted(fd_b.keys()) == sorted(g["bg_feat_dict"].keys())
    for k, v in g["bg_feat_dict"].items():
        assert rel(fd_b[k], v) < (5e-3 if k in loose else 5e-4), f"bg.{k}: {rel(fd_b[k], v):.3e}"
    res = DF.render_train_comp(Pf, frf, Pb, frb, hxy, rng, flow_thresh=meta["flow_thresh"], n_depth=meta["D"])
    for name, ref in (("rendered", g["rendered"]), ("fg", g["aux_fg"]), ("bg", g["aux_bg"])):
        got = res["rendered"] if name == "rendered" else res["aux_dict"][name]
        for k, v in ref.items():
            assert rel(got[k], v) < (5e-3 if k in loose else 5e-4), f"{name}.{k}: {rel(got[k], v):.3e}"
    losses = DF.losses_comp(res, batch, meta["res"], DF.DEFAULT_LOSS_WT)
    for k, v in g["loss"].items():
        assert rel(losses[k], v) < 2e-3, f"loss.{k}: {rel(losses[k], v):.3e}"
    total = sum(losses.values())
    names = list(g["grads"].keys())
    grads = torch.autograd.grad(total, [(Pf if n.startswith("fg:") else Pb)[n[3:]] for n in names], allow_unused=True)
    for n, gv in zip(names, grads):
        ref = g["grads"][n]
        assert gv is not None, n
        e = rel(gv, ref["full"]) if "full" in ref else rel(gv.flatten()[:: ref["stride"]], ref["sub"])
        assert e < 1e-2, f"grad {n}: {e:.3e}"


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
