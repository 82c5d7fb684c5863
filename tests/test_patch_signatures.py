"""lab4d_amd.patch against the REAL reference (build container only; skipped where /root/reference is absent):

  * `patch()` binds into the shimmed reference package and every rebound symbol keeps the reference's signature
    (SURVEY 8b "Python signatures that must not change");
  * every parameter / buffer name the adapters hand to the kernels exists in the reference's own modules
    (a checkpoint loads unchanged);
  * the per-frame inputs the adapters assemble from a reference field have the shapes the kernels expect;
  * `appearance_get_vals` (pure per-frame tensor algebra, runs on CPU) equals the reference's AppearanceEmbedding.get_vals.

No kernel runs here: the device side of the adapters is tests/test_gpu_patch.py."""
import inspect
import sys

import pytest
import torch

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="needs the reference tree (build container only)")


@pytest.fixture(scope="module")
def ns():
    ns = ref_shim.load()
    import importlib
    for m in ("lab4d.nnutils.appearance", "lab4d.nnutils.time", "lab4d.engine.model"):
        importlib.import_module(m)
    return ns


@pytest.fixture()
def patched(ns):
    from lab4d_amd import patch
    saved_q = sys.modules.get("quaternion")
    names = patch.patch(precision="f32", n_depth=64)
    yield patch, names
    patch.unpatch()
    if saved_q is not None:
        sys.modules["quaternion"] = saved_q


def _params(sig):
    return [(p.name, p.kind, p.default) for p in sig.parameters.values()]


def test_every_patched_symbol_keeps_the_reference_signature(ns):
    import importlib
    from lab4d_amd import patch
    checked = 0
    for modname, cls, attr, fn, static in patch.bindings() + patch.INGEST_BINDINGS:
        mod = importlib.import_module(modname)
        owner = mod if cls is None else getattr(mod, cls)
        orig = inspect.getattr_static(owner, attr)  # through the MRO: AppearanceEmbedding.get_vals is TimeMLP's
        if isinstance(orig, staticmethod):
            orig = orig.__func__
        orig = inspect.unwrap(orig)  # decorators like @train_only_fields / @torch.no_grad keep __wrapped__
        so, sn = inspect.signature(orig), inspect.signature(fn)
        if cls is None and attr == "sample_pdf":
            # one extra keyword-only-in-practice trailing argument (return_inds=False); the reference's five come first, unchanged
            assert _params(sn)[: len(so.parameters)] == _params(so), attr
            assert list(sn.parameters)[len(so.parameters):] == ["return_inds"]
        else:
            assert _params(sn) == _params(so), "%s.%s.%s: %s != %s" % (modname, cls, attr, sn, so)
        checked += 1
    assert checked == len(patch.bindings()) + len(patch.INGEST_BINDINGS) >= 32


def test_patch_rebinds_and_unpatch_restores(ns, patched):
    patch, names = patched
    import lab4d.nnutils.nerf as nerf
    import lab4d.nnutils.warping as warping
    import lab4d.engine.model as model
    import lab4d.utils.render_utils as ru
    assert nerf.NeRF.forward is patch.nerf_forward and nerf.NeRF.query_field is patch.query_field
    assert warping.SkinningWarp.forward is patch.skinning_forward and warping.ComposedWarp.forward is patch.composed_forward
    assert model.dvr_model.render is patch.dvr_render and model.dvr_model.evaluate is patch.dvr_evaluate
    from lab4d_amd import render_utils as RU
    assert ru.render_pixel is RU.render_pixel and model.render_pixel is RU.render_pixel and nerf.sample_cam_rays is RU.sample_cam_rays
    import lab4d_amd.quaternion as hq
    assert sys.modules["quaternion"] is hq
    assert "lab4d.engine.model.dvr_model.render" in names and len(names) >= 24
    patch.unpatch()
    assert nerf.NeRF.forward is not patch.nerf_forward and model.render_pixel is not RU.render_pixel


def _fields(ns):
    from lab4d_amd import synthetic
    torch.manual_seed(0)
    out = {}
    for motion in ("skel-quad", "comp_skel-quad_dense"):
        f = ns.deformable.Deformable(motion, ref_shim.synthetic_data_info(64), num_freq_dir=-1, appr_channels=32, num_inst=1, init_scale=0.2)
        f.category = "fg"
        out[motion] = f
    b = ns.nerf.NeRF(ref_shim.synthetic_data_info(64), num_freq_xyz=6, num_freq_dir=0, appr_channels=0, init_scale=0.1)
    b.category = "bg"
    out["bg"] = b
    return out


def test_every_name_the_adapters_read_exists_in_the_reference_modules(ns):
    from lab4d_amd import mlp, patch
    fields = _fields(ns)
    scalars_fg = ["logibeta", "logscale", "logsigma", "aabb", "warp.logibeta", "warp.skinning_model.log_gauss", "warp.skinning_model.symm_idx"]
    nets_fg = [(mlp.NET_FG_BASE, ""), (mlp.NET_FG_COLOR, ""), (mlp.NET_VIS, ""), (mlp.NET_FEAT, ""), (mlp.NET_SKIN, ""), (mlp.NET_SKIN_A, "")]
    for motion in ("skel-quad", "comp_skel-quad_dense"):
        f = fields[motion]
        assert patch.field_kind(f) == "fg"
        P = patch.field_params(f)
        nets = list(nets_fg)
        if motion.startswith("comp_"):
            assert patch.warp_kind(f) == "composed"
            nets += [(mlp.NET_DENSE, "warp.post_warp.forward_map."), (mlp.NET_DENSE, "warp.post_warp.backward_map.")]
        else:
            assert patch.warp_kind(f) == "skinning"
        for net, prefix in nets:
            for b in mlp.bindings(net, prefix):
                assert b.wname in P and b.bname in P, (motion, b.wname)
                assert P[b.wname] is dict(f.named_parameters())[b.wname], "parameters must be the live tensors (autograd reaches them)"
        for k in scalars_fg:
            assert k in P, (motion, k)
        # the warp-level view used by SkinningWarp.forward / ComposedWarp.forward names the same tensors under "warp."
        Pw = patch._warp_params(f.warp)
        for b in mlp.bindings(mlp.NET_SKIN, ""):
            assert Pw[b.wname] is P[b.wname]
    b = fields["bg"]
    assert patch.field_kind(b) == "bg" and patch.warp_kind(b) == "rigid"
    P = patch.field_params(b)
    for net in (mlp.NET_BG_BASE, mlp.NET_BG_COLOR, mlp.NET_VIS):
        for bd in mlp.bindings(net, ""):
            assert bd.wname in P and bd.bname in P, bd.wname
    for k in ("logibeta", "logscale", "aabb"):
        assert k in P


def test_per_frame_inputs_assembled_from_a_reference_field(ns):
    from lab4d_amd import patch, synthetic
    fields = _fields(ns)
    M, N, res = 2, 5, 64
    fr0 = synthetic.make_frames(3, M, res)
    hxy = torch.cat([torch.rand(M, N, 2) * res, torch.ones(M, N, 1)], -1)
    for motion in ("skel-quad", "comp_skel-quad_dense"):
        f = fields[motion]
        f.train()
        sd = {"Kinv": fr0["Kinv"], "field2cam": fr0["field2cam"], "frame_id": torch.tensor([3, 4]), "inst_id": torch.zeros(M, dtype=torch.long),
              "near_far": fr0["near_far"], "hxy": hxy, "feature": torch.randn(M, N, 16)}
        sd["t_articulation"], sd["rest_articulation"] = f.warp.articulation.get_vals_and_mean(sd["frame_id"])
        fr = patch._frames(f, sd)
        B = 25
        for k, shp in {"code_base": (M, 32), "code_color": (M, 32), "code_vis": (M, 32), "code_skin": (M, 32), "appr_code": (M, 32),
                       "t_embed": (M, 128), "t_embed_mean": (1, 128)}.items():
            assert tuple(fr[k].shape) == shp, (motion, k, fr[k].shape)
        assert fr["t_articulation"][0].shape == (M, B, 4) and fr["rest_articulation"][1].shape == (M, B, 4)
        if motion.startswith("comp_"):
            assert tuple(fr["dense"]["t_embed"].shape) == (M, 128) and tuple(fr["dense"]["code_fw"].shape) == (M, 32)
        else:
            assert "dense" not in fr
        # the same values the reference modules produce
        assert torch.equal(fr["t_embed"], f.warp.skinning_model.time_embedding(sd["frame_id"]))
        assert torch.equal(fr["appr_code"], f.appr_embedding.get_vals(sd["frame_id"]))
    # inst_id None -> the mean instance code, one row per frame (base.py:130-134)
    f = fields["skel-quad"]
    c = patch.inst_code(f.basefield, None, 3, torch.device("cpu"))
    assert c.shape == (3, 32) and torch.equal(c[0], f.basefield.inst_embedding.get_mean_embedding())


def test_appearance_get_vals_equals_the_reference(ns):
    from lab4d_amd import patch
    f = _fields(ns)["skel-quad"]
    ae = f.appr_embedding
    with torch.no_grad():
        for p in ae.parameters():  # away from the init (zero biases) so every term counts
            p.add_(0.05 * torch.randn_like(p))
    for fid in (torch.tensor([0, 7, 63]), torch.tensor([31]), None):
        ref = ae.get_vals(fid)
        out = patch.appearance_get_vals(ae, fid)
        assert out.shape == ref.shape
        assert torch.allclose(out, ref, rtol=1e-5, atol=1e-6), float((out - ref).abs().max())
    # gradients reach the reference's own parameters
    g = torch.autograd.grad(patch.appearance_get_vals(ae, torch.tensor([5, 6])).sum(), [ae.output.weight, ae.time_embedding.mapping1.weight])
    gr = torch.autograd.grad(ae.get_vals(torch.tensor([5, 6])).sum(), [ae.output.weight, ae.time_embedding.mapping1.weight])
    for a, b in zip(g, gr):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


def test_per_frame_mlp_adapters_equal_the_reference_modules(ns):
    """Round 6: what patch() binds to TimeEmbedding.forward / CameraMLP.get_vals / IntrinsicsMLP.get_vals, on the REAL reference modules (two videos
    of different length: the per-video normalisation and the InstEmbedding rows matter).  CPU tensors take lab4d_amd.pose's torch algebra -- the same
    algebra the rowmlp programs are held to (tests/test_rowmlp_programs_cpu.py here, tests/test_gpu_rowmlp.py on the MI355X)."""
    import importlib

    import numpy as np
    from lab4d_amd import patch
    pose_ref = importlib.import_module("lab4d.nnutils.pose")
    intr_ref = importlib.import_module("lab4d.nnutils.intrinsics")
    T = 64
    frame_info = {"frame_offset": np.asarray([0, 40, T]), "frame_offset_raw": np.asarray([0, 40, T]), "frame_mapping": list(range(T))}
    torch.manual_seed(2)
    cam = pose_ref.CameraMLP(ref_shim.synthetic_data_info(T)["rtmat"], frame_info=frame_info, W=64)
    intr = intr_ref.IntrinsicsMLP(np.tile(np.asarray([[64.0, 64.0, 32.0, 32.0]], dtype=np.float32), (T, 1)), frame_info=frame_info, W=64)
    with torch.no_grad():
        cam.base_quat.copy_(torch.randn(2, 4))
        intr.base_logfocal.copy_(torch.tensor([[4.1, 4.2], [4.0, 3.9]]))
        for m in (cam, intr):
            for q in m.parameters():
                q.add_(0.03 * torch.randn_like(q))
    for fid in (torch.tensor([3, 39, 40, 63]), None):
        q0, t0 = cam.get_vals(fid)
        q1, t1 = patch.camera_get_vals(cam, fid)
        assert torch.allclose(q1, q0, rtol=1e-5, atol=1e-6) and torch.allclose(t1, t0, rtol=1e-5, atol=1e-6)
        assert torch.allclose(patch.intrinsics_get_vals(intr, fid), intr.get_vals(fid), rtol=1e-5, atol=1e-5)
        te0 = cam.time_embedding(fid)
        assert torch.allclose(patch.time_embedding_forward(cam.time_embedding, fid), te0, rtol=1e-5, atol=1e-6)
    fid = torch.tensor([5, 41])
    assert torch.allclose(patch.time_embedding_forward(cam.time_embedding, fid[:, None]), cam.time_embedding(fid[:, None]), rtol=1e-5, atol=1e-6)
    leaves = [cam.base_quat, cam.quat[2].weight, cam.linear_1[0].weight, cam.time_embedding.mapping1.weight, cam.time_embedding.inst_embedding.mapping.weight]
    ga = torch.autograd.grad(sum(x.sin().sum() for x in patch.camera_get_vals(cam, fid)), leaves)
    gb = torch.autograd.grad(sum(x.sin().sum() for x in cam.get_vals(fid)), leaves)
    for a, b in zip(ga, gb):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


def test_draw_rng_follows_the_reference_draw_order():
    """multinomial (nerf.py:437-440) first, then randperm (feature.py:177): same generator state -> same draws."""
    from lab4d_amd import patch
    M, N, D = 2, 40, 8
    torch.manual_seed(7)
    eik = torch.multinomial(torch.ones(M * N), M * N // 16, replacement=False)
    perm = torch.randperm(M * N * D)[:1024]
    torch.manual_seed(7)
    r = patch.draw_rng(M, N, D, torch.device("cpu"))
    assert torch.equal(r["eik_inds"], eik) and torch.equal(r["match_perm"], perm)


def test_optimizer_init_adopts_the_reference_optimizer(ns, patched):
    """Trainer.optimizer_init through the binding: the reference's own parameter / learning-rate selection and OneCycleLR run unchanged, the
    AdamW they built becomes a TorchFlatAdamW IN PLACE (same object, same param_groups the scheduler writes to), every parameter's .data and
    .grad are views of the flat buffers, and state_dict() / load_state_dict() round-trip (the two-rounds-back cache of check_grad)."""
    import types
    import importlib
    from lab4d_amd import synthetic
    from lab4d_amd.optim import TorchFlatAdamW
    sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden"))
    import make_golden as MG
    patch, _ = patched
    trainer_mod = importlib.import_module("lab4d.engine.trainer")
    field = MG.build_reference_field(ns, synthetic.make_weights(5))
    model = torch.nn.Module()
    model.module = torch.nn.Module()
    model.module.fields = torch.nn.Module()
    model.module.fields.field_params = torch.nn.ModuleDict({"fg": field})
    t = types.SimpleNamespace(opts={"learning_rate": 5e-4, "freeze_bone_len": False, "num_rounds": 20}, model=model, total_steps=400)
    t.get_lr_dict = types.MethodType(trainer_mod.Trainer.get_lr_dict, t)
    trainable = [p for p in model.parameters() if p.requires_grad]
    before = [p.detach().clone() for p in trainable]
    trainer_mod.Trainer.optimizer_init(t)
    assert isinstance(t.optimizer, TorchFlatAdamW) and t.scheduler.optimizer is t.optimizer
    lrs = {id(g["params"][0]): g["lr"] for g in t.optimizer.param_groups}
    named = dict(model.named_parameters())
    # the reference's rule (trainer.py:122-148): 10x the base rate for the explicitly listed scalars, OneCycleLR starts at 1/25 of it
    assert abs(lrs[id(named["module.fields.field_params.fg.logibeta"])] - 5e-3 / 25) < 1e-12
    assert abs(lrs[id(named["module.fields.field_params.fg.basefield.linear_1.0.weight"])] - 5e-4 / 25) < 1e-12
    flat = t.optimizer.flat
    for p, b in zip(trainable, before):
        assert torch.equal(p.detach(), b)
    inside = [p for g in t.optimizer.param_groups for p in g["params"]]
    lo, hi = flat.flat.data_ptr(), flat.flat.data_ptr() + 4 * flat.n
    assert all(lo <= p.data_ptr() < hi for p in inside) and all(p.grad is not None for p in inside)
    # optimizer.step() must reach TorchFlatAdamW.step: OneCycleLR wrapped the bound AdamW.step as an INSTANCE attribute before the class swap, and
    # an instance attribute outlives it (the flat kernel itself needs the device: the call is recorded instead)
    calls = []
    flat.step = lambda *a, **k: calls.append("flat.step")
    t.optimizer.step()
    del flat.step
    assert calls == ["flat.step"] and t.optimizer._opt_called
    t.scheduler.step()
    assert t.optimizer._lrs() != t.optimizer._lr_sent  # OneCycleLR wrote the groups; the next step() uploads the new rates
    from copy import deepcopy
    sd = deepcopy(t.optimizer.state_dict())  # what Trainer.save_checkpoint caches (trainer.py:262-269)
    flat.m.fill_(3.0)
    t.optimizer.load_state_dict(sd)
    assert all(float(t.optimizer.state[p]["exp_avg"].abs().max()) == 0.0 for p in inside)
    assert t.optimizer.state[inside[0]]["exp_avg"].data_ptr() == flat.m.data_ptr()  # still views of the flat moment buffer


def test_proxy_refresh_and_ingestion_bindings(ns):
    """Round 4 (SURVEY 8f rows 3-4 through patch()): NeRF.extract_canonical_mesh / update_aabb / update_near_far are rebound by default,
    VidDataset.load_data only with patch(ingest=True); every attribute the adapters read exists on the real reference classes."""
    import importlib
    from lab4d_amd import patch
    saved_q = sys.modules.get("quaternion")
    try:
        names = patch.patch(precision="f32", ingest=True)
        nerf = importlib.import_module("lab4d.nnutils.nerf")
        vid = importlib.import_module("lab4d.dataloader.vidloader")
        assert nerf.NeRF.extract_canonical_mesh is patch.nerf_extract_canonical_mesh and nerf.NeRF.update_aabb is patch.nerf_update_aabb
        assert nerf.NeRF.update_near_far is patch.nerf_update_near_far and vid.VidDataset.load_data is patch.vid_load_data
        assert "lab4d.dataloader.vidloader.VidDataset.load_data" in names
    finally:
        patch.unpatch()
        if saved_q is not None:
            sys.modules["quaternion"] = saved_q
    vid = importlib.import_module("lab4d.dataloader.vidloader")
    assert vid.VidDataset.load_data is not patch.vid_load_data
    patch.patch(precision="f32")
    try:
        assert vid.VidDataset.load_data is not patch.vid_load_data  # opt-in only
    finally:
        patch.unpatch()
        if saved_q is not None:
            sys.modules["quaternion"] = saved_q
    # names the adapters read on the reference objects
    src = inspect.getsource(vid.VidDataset)
    for attr in ("mmap_list", "crop2raw", "is_detected", "dataid", "frame_info", "delta_list", "pixels_per_image", "load_pair", "sample_delta"):
        assert "self." + attr in src, attr
    nsrc = inspect.getsource(importlib.import_module("lab4d.nnutils.nerf").NeRF)
    for attr in ("proxy_geometry", "aabb", "near_far", "camera_mlp", "vis_mlp", "basefield", "pos_embedding", "category"):
        assert "self." + attr in nsrc, attr
    geom = importlib.import_module("lab4d.utils.geom_utils")
    assert list(inspect.signature(geom.marching_cubes).parameters)[:2] == ["sdf_func", "aabb"]
    # the served-volume closures walk the grid in eval_func_chunk's order: one call per chunk, rows consecutive
    import torch
    vol = torch.arange(10.0).reshape(-1, 1)
    pos = [0]

    def f(xyz):
        out = vol[pos[0]:pos[0] + xyz.shape[0]]
        pos[0] += xyz.shape[0]
        return out
    assert torch.equal(geom.eval_func_chunk(f, torch.zeros(10, 3), chunk_size=4), vol)
