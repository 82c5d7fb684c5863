"""lab4d_amd.patch on the device: the adapters (same signatures as the reference's methods) driven with stand-in module objects
(tests/standins.py: same attribute names, parameters under the reference's state_dict names, per-frame modules replaced by
the rows the reference produced) against the reference-generated fixtures.  The build-container half -- binding into the real
reference, signature equality, state_dict names -- is tests/test_patch_signatures.py."""
import os
import types

import pytest
import torch

from lab4d_amd import synthetic
from oracle import lab4d_oracle as O
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import standins  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture()
def patch_f32():
    from lab4d_amd import mlp, patch
    saved = (patch.PRECISION, patch.N_DEPTH, patch.draw_rng)
    patch.PRECISION = mlp.PREC_F32
    yield patch
    patch.PRECISION, patch.N_DEPTH, patch.draw_rng = saved


def _load(golden_dir, name):
    g = torch.load(os.path.join(golden_dir, name), weights_only=False)
    meta = g["meta"]
    motion = meta.get("fg_motion", "skel-quad")
    P = synthetic.make_weights(meta["seed"], num_inst=meta.get("num_inst", 1), sdf_bias=meta.get("sdf_bias"),
                               num_bones=18 if "skel-human" in motion else 25, motion=motion if motion in ("rigid", "dense") else "skinning")
    composed = meta.get("fg_motion", "skel-quad").startswith("comp_")
    if composed:
        P = synthetic.add_dense_weights(P, meta["seed"], meta.get("num_inst", 1))
    return g, synthetic.to_device(P, DEV), composed


def _samples_dict(g, with_feature=True):
    fr = synthetic.to_device(dict(g["frames"]), DEV)
    sd = {k: fr[k] for k in ("Kinv", "field2cam", "frame_id", "inst_id", "near_far", "t_articulation", "rest_articulation")}
    if g["meta"].get("fg_motion") in ("rigid", "dense"):  # only SkinningWarp fields put articulations into the samples (deformable.py:254-289)
        del sd["t_articulation"], sd["rest_articulation"]
    sd["hxy"] = g["hxy"].to(DEV)
    if with_feature:
        sd["feature"] = g["batch"]["feature"].to(DEV)
    return fr, sd


def _model(field):
    from lab4d_amd import patch
    m = types.SimpleNamespace(fields=types.SimpleNamespace(field_params={"fg": field}))
    m.render_samples = types.MethodType(patch.dvr_render_samples, m)
    m.render_samples_chunk = types.MethodType(patch.dvr_render_samples_chunk, m)
    return m


@pytest.mark.parametrize("case", ["train_small.pt", "train_compmotion.pt", "train_human.pt", "train_rigid.pt", "train_dense.pt"])
def test_query_field_and_render_samples_match_the_reference(golden_dir, patch_f32, case):
    """Deformable.query_field -> dvr_model.render_samples_chunk through the adapters, training mode, vs the reference's own
    outputs for the same rays / weights / random draws; gradients reach the stand-in module's parameters."""
    from lab4d_amd import deformable as DF
    patch = patch_f32
    g, P, composed = _load(golden_dir, case)
    meta = g["meta"]
    fr, sd = _samples_dict(g)
    field = standins.fg_field(P, fr, composed=composed, alpha=meta["alpha"], training=True, motion=meta.get("fg_motion"))
    patch.N_DEPTH = meta["D"]
    rng = synthetic.to_device(g["rng"], DEV)
    patch.draw_rng = lambda M, N, D, device: rng  # the fixture's draws instead of fresh ones
    fd, deltas, aux = patch.query_field(field, sd, flow_thresh=meta["flow_thresh"])
    assert set(fd.keys()) == set(g["feat_dict"].keys())
    for k, v in g["feat_dict"].items():
        assert rel(fd[k], v) < 2e-4, (k, rel(fd[k], v))
    assert rel(deltas, g["deltas"]) < 1e-5
    res = patch.dvr_render_samples_chunk(_model(field), {"fg": sd}, flow_thresh=meta["flow_thresh"], chunk_size=8192)
    assert set(res["rendered"].keys()) == set(g["rendered"].keys())
    for k, v in g["rendered"].items():
        assert rel(res["rendered"][k], v) < 2e-4, (k, rel(res["rendered"][k], v))
    for k, v in g["aux_fg"].items():
        assert rel(res["aux_dict"]["fg"][k], v) < 2e-4, (k, rel(res["aux_dict"]["fg"][k], v))
    losses = DF.losses_fg(res, synthetic.to_device(g["batch"], DEV), meta["res"], DF.DEFAULT_LOSS_WT)
    params = dict(field.named_parameters())
    names = [k for k in g["grads"] if not k.startswith("frame:") and k in params]
    assert len(names) > 40
    grads = torch.autograd.grad(sum(v for v in losses.values() if bool(torch.isfinite(v))), [params[k] for k in names], allow_unused=True)
    for k, gv in zip(names, grads):
        ref = g["grads"][k]
        assert gv is not None, k
        e = rel(gv, ref["full"]) if "full" in ref else rel(gv.flatten()[:: ref["stride"]], ref["sub"])
        assert e < 5e-3, (k, e)


@pytest.mark.parametrize("case", ["eval_small.pt", "eval_rigid.pt", "eval_dense.pt"])
def test_eval_query_field_and_chunking(golden_dir, patch_f32, case):
    """Eval mode (importance sampling, valid-sample compaction, normals) through the adapters vs the reference fixture; chunked
    rendering concatenates to the unchunked result (per-ray quantities; "vis" is normalised per chunk by design).
    eval_rigid / eval_dense: fg_motion "rigid" / "dense" (no articulations in the samples: the valid mask is the aabb test alone)."""
    patch = patch_f32
    g, P, _ = _load(golden_dir, case)
    fr, sd = _samples_dict(g, with_feature=False)
    field = standins.fg_field(P, fr, training=False, motion=g["meta"].get("fg_motion"))
    patch.N_DEPTH = g["meta"]["D"]
    fd, deltas, aux = patch.query_field(field, sd)
    assert aux == {}
    for k, v in g["feat_dict"].items():
        # per-sample normal / eikonal: a normalised (squared) first derivative, see RENDER_TOL_F32 in test_gpu_field.py
        assert rel(fd[k], v) < (5e-3 if k in ("normal", "eikonal") else 5e-4), (k, rel(fd[k], v))
    one = patch.dvr_render_samples_chunk(_model(field), {"fg": sd})
    for k, v in g["rendered"].items():
        assert rel(one["rendered"][k], v) < 5e-4, (k, rel(one["rendered"][k], v))
    M, N = sd["hxy"].shape[:2]
    many = patch.dvr_render_samples_chunk(_model(field), {"fg": sd}, chunk_size=M * 3)  # 3 pixels per frame and chunk
    for k, v in one["rendered"].items():
        assert many["rendered"][k].shape == v.shape
        if k != "vis":
            assert rel(many["rendered"][k], v.cpu()) < 1e-5, k


def test_module_forwards(golden_dir, patch_f32):
    """NeRF.forward / VisField.forward / FeatureNeRF.compute_feat / SkinningWarp.forward / backward_warp / forward_warp through
    the adapters vs the oracle on the same weights; incl. the (K,1,1,3) per-sample-row form the reference's compacted eval
    path uses (nerf.py:795-798) and inst_id=None (mean instance code)."""
    patch = patch_f32
    g, P, _ = _load(golden_dir, "train_small.pt")
    fr, sd = _samples_dict(g)
    field = standins.fg_field(P, fr, training=True)
    Pc = {k: v.detach().cpu() for k, v in P.items()}
    frc = synthetic.add_codes(dict(g["frames"]), Pc)
    gen = torch.Generator().manual_seed(3)
    M, N, D = 2, 5, 7
    xyz = torch.randn(M, N, D, 3, generator=gen) * 0.08
    fid, iid = fr["frame_id"], fr["inst_id"]
    rgb, dens = patch.nerf_forward(field, xyz.to(DEV), dir=xyz.to(DEV), frame_id=fid, inst_id=iid)
    rgb_o, dens_o = O.nerf_forward(Pc, xyz, {"basefield": frc["code_base"], "colorfield": frc["code_color"]}, appr_code=frc["appr_code"])
    assert rel(rgb, rgb_o) < 2e-4 and rel(dens, dens_o) < 2e-4
    sdf = patch.nerf_forward(field, xyz.to(DEV), inst_id=iid, get_density=False)
    assert rel(sdf, O.nerf_forward(Pc, xyz, {"basefield": frc["code_base"]}, with_color=False, get_density=False)) < 2e-4
    # compacted form: K samples as (K,1,1,3) with per-sample frame / instance ids == the (M,N,D) call, gathered
    flat = xyz.reshape(-1, 1, 1, 3).to(DEV)
    fid_s = fid[:, None, None].expand(M, N, D).reshape(-1)
    iid_s = iid[:, None, None].expand(M, N, D).reshape(-1)
    rgb_k, dens_k = patch.nerf_forward(field, flat, dir=flat, frame_id=fid_s, inst_id=iid_s)
    assert rel(rgb_k.reshape(M, N, D, 3), rgb.cpu()) < 1e-5 and rel(dens_k.reshape(M, N, D, 1), dens.cpu()) < 1e-5
    # mean instance
    sdf_m = patch.nerf_forward(field, xyz.to(DEV), inst_id=None, get_density=False)
    mean_code = Pc["basefield.inst_embedding.mapping.weight"].mean(0, keepdim=True).expand(M, -1)
    assert rel(sdf_m, O.nerf_forward(Pc, xyz, {"basefield": mean_code}, with_color=False, get_density=False)) < 2e-4
    assert rel(patch.vis_forward(field.vis_mlp, xyz.to(DEV), inst_id=iid), O.vis_field(Pc, xyz, frc["code_vis"])) < 2e-4
    assert rel(patch.compute_feat(field, xyz.to(DEV))["feature"], O.compute_feat(Pc, xyz)) < 2e-4
    field.eval()
    assert patch.compute_feat(field, xyz.to(DEV)) == {}  # train-only field (decorator.py:4-17)
    field.train()
    sdict = {"t_articulation": fr["t_articulation"], "rest_articulation": fr["rest_articulation"]}
    for backward in (True, False):
        out, aux = patch.skinning_forward(field.warp, xyz.to(DEV), fid, iid, backward=backward, samples_dict=sdict, return_aux=True)
        te = frc["t_embed"] if backward else frc["t_embed_mean"]
        out_o, aux_o = O.skinning_warp(Pc, xyz, frc["t_articulation"], frc["rest_articulation"], te, frc["code_skin"], backward)
        assert rel(out, out_o) < 2e-4
        for k in aux_o:
            assert rel(aux[k], aux_o[k]) < 5e-4, k
    # without articulations in samples_dict the warp asks its articulation module (warping.py:299-304)
    out2 = patch.skinning_forward(field.warp, xyz.to(DEV), fid, iid, backward=True)
    assert rel(out2, O.skinning_warp(Pc, xyz, frc["t_articulation"], frc["rest_articulation"], frc["t_embed"], frc["code_skin"], True)[0]) < 2e-4
    # Deformable.backward_warp / forward_warp
    xyz_cam = torch.randn(M, N, D, 3, generator=gen) * 0.05 + torch.tensor([0.0, 0.0, 0.6])
    field.warp.forward = types.MethodType(patch.skinning_forward, field.warp)
    bw = patch.backward_warp(field, xyz_cam.to(DEV), xyz_cam.to(DEV), fr["field2cam"], fid, iid, samples_dict=sdict)
    xt_o, dir_o = O.cam_to_field(xyz_cam, xyz_cam, frc["field2cam"])
    x_o, aux_o = O.skinning_warp(Pc, xt_o, frc["t_articulation"], frc["rest_articulation"], frc["t_embed"], frc["code_skin"], True)
    assert rel(bw["xyz_t"], xt_o) < 1e-5 and rel(bw["dir"], dir_o) < 1e-5 and rel(bw["xyz"], x_o) < 2e-4
    assert set(bw.keys()) == {"xyz", "dir", "xyz_t", "skin_entropy", "delta_skin"}
    fw = patch.forward_warp(field, xyz.to(DEV), fr["field2cam"], fid, iid, samples_dict=sdict)
    xf_o, _ = O.skinning_warp(Pc, xyz, frc["t_articulation"], frc["rest_articulation"], frc["t_embed_mean"], frc["code_skin"], False)
    assert rel(fw, O.field_to_cam(xf_o, frc["field2cam"])) < 2e-4


def test_composed_and_dense_forward_match_the_reference(golden_dir, patch_f32):
    """ComposedWarp.forward / DenseWarp.forward through the adapters vs the reference-generated comp_warp.pt."""
    patch = patch_f32
    g = torch.load(os.path.join(golden_dir, "comp_warp.pt"), weights_only=False)
    P = synthetic.to_device(synthetic.add_dense_weights(synthetic.make_weights(0)), DEV)
    fr = synthetic.to_device(dict(g["frames"]), DEV)
    field = standins.fg_field(P, fr, composed=True)
    xyz = g["xyz"].to(DEV)
    fid, iid = fr["frame_id"], fr["inst_id"]
    sdict = {"t_articulation": fr["t_articulation"], "rest_articulation": fr["rest_articulation"]}
    out_bw, aux = patch.composed_forward(field.warp, xyz, fid, iid, backward=True, samples_dict=sdict, return_aux=True)
    assert rel(out_bw, g["out_bw"]) < 2e-4
    for k, v in g["aux_bw"].items():
        assert rel(aux[k], v) < 5e-4, k
    assert rel(patch.composed_forward(field.warp, xyz, fid, iid, samples_dict=sdict), g["out_fw"]) < 2e-4
    assert rel(patch.composed_forward(field.warp, xyz, None, iid, samples_dict=sdict), g["out_fw_none"]) < 2e-4
    assert rel(patch.dense_forward(field.warp.post_warp, xyz, fid, iid, backward=False), g["dense_fw"]) < 2e-4
    assert rel(patch.dense_forward(field.warp.post_warp, xyz, fid, iid, backward=True), g["dense_bw"]) < 2e-4


def test_evaluate_and_render_entry_points(golden_dir, patch_f32):
    """dvr_model.render / evaluate through the adapters on a stand-in model: a 4x4-pixel frame pair, eval mode.  evaluate()
    returns (frames, H, W, c) images of the first frame of every pair, masked by the rendered mask (model.py:197-209)."""
    patch = patch_f32
    g, P, _ = _load(golden_dir, "eval_small.pt")
    fr, _ = _samples_dict(g, with_feature=False)
    field = standins.fg_field(P, fr, training=False)
    patch.N_DEPTH = 16
    res, M = 4, 2
    hxy = synthetic.make_rays(res, M).to(DEV) * torch.tensor([16.0, 16.0, 1.0], device=DEV)  # spread over the 64-pixel image plane
    m = _model(field)
    m.process_frameid = lambda batch: None
    m.get_samples = lambda batch: {"fg": dict({k: fr[k][batch["idx"]] if not isinstance(fr[k], tuple) else tuple(t[batch["idx"]] for t in fr[k])
                                               for k in ("Kinv", "field2cam", "frame_id", "inst_id", "near_far", "t_articulation",
                                                         "rest_articulation")}, hxy=batch["hxy"])}
    m.render = types.MethodType(patch.dvr_render, m)
    batch = {"frameid": fr["frame_id"], "idx": torch.arange(M, device=DEV), "hxy": hxy}
    out = patch.dvr_render(m, batch)
    assert out["rendered"]["rgb"].shape == (M, res * res, 3)
    ev = patch.dvr_evaluate(m, batch, is_pair=True)
    assert ev["rgb"].shape == (1, res, res, 3) and ev["mask"].shape == (1, res, res, 1)
    raw = out["rendered"]
    assert torch.allclose(ev["rgb"][0].reshape(-1, 3), (raw["rgb"] * raw["mask"])[0], atol=1e-6)
    assert torch.allclose(ev["mask"][0].reshape(-1, 1), raw["mask"][0], atol=1e-6)
    ev1 = patch.dvr_evaluate(m, batch, is_pair=False)
    assert ev1["rgb"].shape == (M, res, res, 3)


def test_compute_loss_and_optimizer_bindings(golden_dir, patch_f32):
    """dvr_model.compute_loss, Trainer.check_grad and the optimizer step through the adapters on stand-in objects: the loss_dict against the
    reference's own values for the fixture (keys, order, numbers), then check_grad + optimizer.step() against torch's clip_grad_norm_ +
    AdamW on a copy of the parameters, incl. a blown-up step that must be discarded without touching weights, moments or step count."""
    from lab4d_amd import deformable as DF
    from lab4d_amd.optim import TorchFlatAdamW
    patch = patch_f32
    g, P, composed = _load(golden_dir, "train_small.pt")
    meta = g["meta"]
    fr, sd = _samples_dict(g)
    field = standins.fg_field(P, fr, composed=composed, alpha=meta["alpha"], training=True)
    patch.N_DEPTH = meta["D"]
    rng = synthetic.to_device(g["rng"], DEV)
    patch.draw_rng = lambda M, N, D, device: rng
    model = _model(field)
    model.config = dict(DF.DEFAULT_LOSS_WT, field_type="fg", train_res=meta["res"])
    reg_vals = {"reg_visibility": torch.tensor(0.3, device=DEV), "reg_soft_deform": torch.tensor(0.0, device=DEV),
                "reg_gauss_skin": torch.tensor(0.2, device=DEV), "reg_cam_prior": torch.tensor(0.5, device=DEV), "reg_skel_prior": torch.tensor(0.1, device=DEV)}

    def compute_reg_loss(loss_dict, results):  # what the reference's method adds (model.py:503-526): field-level terms + the rendered ones
        loss_dict.update(reg_vals)
        loss_dict["reg_eikonal"] = results["rendered"]["eikonal"]

    def apply_loss_weights(loss_dict, config):  # model.py:587-611
        for k, v in loss_dict.items():
            loss_dict[k] = v[v > 0].mean() * config.get(k + "_wt", 1.0)

    model.compute_reg_loss, model.apply_loss_weights = compute_reg_loss, apply_loss_weights
    batch = synthetic.to_device(g["batch"], DEV)
    params = [p for p in field.parameters()]
    opt = TorchFlatAdamW([{"params": [p], "lr": 1e-3} for p in params], lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4)
    ref_params = [p.detach().clone().requires_grad_(True) for p in params]
    ref_opt = torch.optim.AdamW([{"params": [p], "lr": 1e-3} for p in ref_params], lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4)
    trainer = types.SimpleNamespace(optimizer=opt, model_cache=[None, None], optimizer_cache=[None, None], scheduler_cache=[None, None])
    for it, scale in enumerate([1.0, 1e5, 1.0]):
        opt.zero_grad()
        res = patch.dvr_render_samples_chunk(model, {"fg": sd}, flow_thresh=meta["flow_thresh"], chunk_size=8192)
        loss_dict = patch.dvr_compute_loss(model, batch, res)
        if it == 0:
            want = [k for k in patch.LOSS_ORDER if k in g["loss"] or k in reg_vals]
            assert list(loss_dict) == want, (list(loss_dict), want)
            for k, v in g["loss"].items():
                assert rel(loss_dict[k], v) < 5e-4, (k, rel(loss_dict[k], v))
            for k, v in reg_vals.items():  # a zero term is the mean of an empty selection = NaN, like the reference
                w = DF.DEFAULT_LOSS_WT[k + "_wt"]
                assert (bool(torch.isnan(loss_dict[k])) if float(v) == 0 else abs(float(loss_dict[k]) - float(v) * w) < 1e-7), k
        total = torch.sum(torch.stack([v for v in loss_dict.values() if bool(torch.isfinite(v))])) * scale
        total.backward()
        for p, q in zip(params, ref_params):
            q.grad = p.grad.detach().clone()
        before = opt.flat.flat.clone()
        patch.trainer_check_grad(trainer, thresh=5.0)
        opt.step()
        tn = torch.nn.utils.clip_grad_norm_(ref_params, 5.0)
        if float(tn) > 5.0:
            ref_opt.zero_grad()
        ref_opt.step()
        torch.cuda.synchronize()
        assert int(opt.skipped) == int(scale > 1), (it, float(tn))
        if scale > 1:
            assert torch.equal(opt.flat.flat, before)
        for p, q in zip(params, ref_params):
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), it
    assert int(opt.flat.dev_step) == 2
    sd_opt = opt.state_dict()
    assert float(sd_opt["state"][0]["step"]) == 2.0


def test_appearance_get_vals_on_the_device(golden_dir, patch_f32):
    """AppearanceEmbedding.get_vals through the adapter on the device: a stand-in module holding the reference module's own
    state_dict (tests/golden/pose.pt "appr_state": randomised AppearanceEmbedding(frame_info, 32)) and its frame tables, against the values
    and parameter gradients the reference produced for the same frame ids -- and for frame_id=None (all frames)."""
    patch = patch_f32
    fx = torch.load(os.path.join(golden_dir, "pose.pt"), weights_only=False)
    ti, at = fx["time_info"], fx["appr_time"]
    ae = standins.Node()
    standins._tree(ae, {k: v.to(DEV) for k, v in fx["appr_state"].items()})
    te = ae.time_embedding
    for k in ("frame_to_vid", "frame_mapping", "raw_fid_to_vid", "raw_fid_to_vidlen", "raw_fid_to_vstart"):
        setattr(te, k, ti[k].to(DEV))
    te.fourier_embedding = types.SimpleNamespace(N_freqs=at["num_freq_t"])

    def frame_to_tid(frame_id):  # embedding.py:179-187
        fid = frame_id.long()
        return (frame_id - te.raw_fid_to_vstart[fid] - te.raw_fid_to_vidlen[fid] / 2) / at["max_ts"] * 2 * at["time_scale"]

    te.frame_to_tid = frame_to_tid
    fid = fx["frame_id"].to(DEV)
    ref = fx["appr"]
    out = patch.appearance_get_vals(ae, fid)
    assert out.is_cuda and rel(out, ref["vals"]) < 1e-5, rel(out, ref["vals"])
    assert rel(patch.appearance_get_vals(ae, None), ref["all_frames"]) < 1e-5
    (out * ref["cot"].to(DEV)).sum().backward()
    params = dict(ae.named_parameters())
    assert set(ref["grads"]) <= set(params)
    for k, g in ref["grads"].items():
        assert rel(params[k].grad, g) < 1e-4, (k, rel(params[k].grad, g))
