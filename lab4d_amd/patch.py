"""Operator-API adapter: the reference's own method signatures, bodies on the HIP library.

`patch()` rebinds, inside an importable `lab4d` package, every entry point SURVEY.md 8b lists as "must not change":

    quaternion (package)                      third_party/quaternion/__init__.py:2-3      -> lab4d_amd.quaternion
    render_utils.sample_cam_rays / render_pixel / compute_weights / integrate / sample_pdf
                                              utils/render_utils.py:8,59,99,129,187       -> lab4d_amd.render_utils
    NeRF.forward                              nnutils/nerf.py:167                          -> nerf_forward
    NeRF.query_field (+ FeatureNeRF / Deformable overrides)
                                              nnutils/nerf.py:580, feature.py:89, deformable.py:300   -> query_field
    NeRF.backward_warp / forward_warp (+ Deformable overrides)
                                              nnutils/nerf.py:865,889, deformable.py:119,154           -> backward_warp / forward_warp
    VisField.forward                          nnutils/visibility.py:53                     -> vis_forward
    FeatureNeRF.compute_feat                  nnutils/feature.py:136                       -> compute_feat
    SkinningWarp.forward                      nnutils/warping.py:277                       -> skinning_forward
    DenseWarp.forward                         nnutils/warping.py:143                       -> dense_forward
    ComposedWarp.forward                      nnutils/warping.py:445                       -> composed_forward
    MultiFields.compose_fields                nnutils/multifields.py:339                   -> compose_fields
    AppearanceEmbedding.get_vals              nnutils/appearance.py:8-56, time.py:107      -> appearance_get_vals
    TimeEmbedding.forward                     nnutils/embedding.py:194-217                 -> time_embedding_forward   (rowmlp program: one launch)
    CameraMLP.get_vals                        nnutils/pose.py:130-150                      -> camera_get_vals          (rowmlp program)
    IntrinsicsMLP.get_vals                    nnutils/intrinsics.py:86-107                 -> intrinsics_get_vals      (rowmlp program)
    dvr_model.render / evaluate / render_samples / render_samples_chunk
                                              engine/model.py:162,217,259,328              -> dvr_*
    dvr_model.compute_loss                    engine/model.py:375-399 (+ 401-611)          -> dvr_compute_loss (fused per-ray loss kernels)
    Trainer.optimizer_init / check_grad       engine/trainer.py:150-210,581-604            -> trainer_* (TorchFlatAdamW, device-side discard rule)

Every function below takes the reference module as `self` and reads only attributes the reference defines (parameter
names = the reference's state_dict names: a checkpoint loads unchanged); per-frame quantities (instance / time /
appearance codes, articulations) come from the reference's own per-frame modules hanging off `self`, per-sample work
goes to liblab4d_hip.so.  There is no fallback: a field shape without a kernel instantiation raises NotImplementedError,
a CPU tensor raises RuntimeError.

Tests: tests/test_patch_signatures.py (build container: binds into the real reference, `inspect.signature` equality for
every patched symbol, every state_dict key the adapters read exists in the reference's modules) and
tests/test_gpu_patch.py (MI355X: the adapters driven with stand-in module objects against the reference-generated
fixtures).
"""
import math
import sys
from collections import defaultdict

import torch

from . import deformable as DF
from . import mlp, multifields
from . import render_utils as RU
from . import warping as W

# Process-wide DEFAULTS, set by patch(precision=..., n_depth=...); a model overrides them for itself with configure(model, ...), which
# stores the setting on the module objects -- two models with different settings can live in one process.
# precision of the MLP chains the adapters launch (mlp.PREC_BF16 = BASELINE configs[1]; mlp.PREC_F32 = 1e-4 parity path)
PRECISION = mlp.PREC_BF16
# samples per ray: the reference hard-codes the defaults of sample_cam_rays (render_utils.py:8) and importance_sampling
# (nerf.py:686-696), n_depth=64; BASELINE configs[1] / [4] ask for 128 / 256
N_DEPTH = 64
_PREC_NAMES = {"bf16": mlp.PREC_BF16, "f32": mlp.PREC_F32}


def configure(module, precision=None, n_depth=None):
    """Per-model settings: every sub-module of `module` (a dvr_model, a MultiFields, one field, one warp ...) runs its kernels at this
    precision ("bf16" / "f32") / with this many samples per ray, whatever the process-wide defaults of patch() are."""
    for m in module.modules():
        if precision is not None:
            m._lab4d_amd_precision = _PREC_NAMES[precision] if isinstance(precision, str) else int(precision)
        if n_depth is not None:
            m._lab4d_amd_n_depth = int(n_depth)
    return module


def _prec(self):
    return getattr(self, "_lab4d_amd_precision", PRECISION)


def _ndepth(self):
    return getattr(self, "_lab4d_amd_n_depth", N_DEPTH)


# ---------------------------------------------------------------------------------------------------
# reading the reference modules
# ---------------------------------------------------------------------------------------------------
def params_of(module, prefix=""):
    """Flat {state_dict name: tensor} of a reference module: parameters (live, so autograd reaches them) + buffers."""
    P = {prefix + k: v for k, v in module.named_parameters()}
    for k, v in module.named_buffers():
        P.setdefault(prefix + k, v)
    return P


def field_params(field):
    """Parameters of a NeRF / Deformable field under the names lab4d_amd's functional code uses (= state_dict names), plus
    the non-parameter attributes it needs (skinning symmetry table: a plain list on SkinningField, skinning.py:88)."""
    P = params_of(field)
    warp = getattr(field, "warp", None)
    sk = getattr(warp, "skinning_model", None)
    if sk is not None and getattr(sk, "symm_idx", None) is not None:
        P["warp.skinning_model.symm_idx"] = torch.as_tensor(sk.symm_idx, dtype=torch.long, device=P["logibeta"].device)
    return P


def inst_code(cond_mlp, inst_id, rows, device):
    """CondMLP's instance code for `rows` leading rows (base.py:123-146): inst_id None -> the mean embedding."""
    emb = cond_mlp.inst_embedding
    if inst_id is None:
        return emb.get_mean_embedding().to(device).reshape(1, -1).expand(rows, -1)
    return emb(inst_id)


def _spf(x):
    n = 1
    for d in x.shape[1:-1]:
        n *= d
    return n


def field_kind(field):
    """Which kernel instantiation serves this NeRF: "fg" (multifields.py:77-84: W=256, D=8, 10 xyz frequencies, no view
    direction, 32-channel appearance code) or "bg" (multifields.py:86-93: W=128, D=5, 6 frequencies, raw view direction)."""
    nf = field.pos_embedding.N_freqs
    w = field.sdf.in_features
    dir_c = field.dir_embedding.out_channels
    if nf == 10 and w == 256 and dir_c == 0 and field.appr_channels == 32:
        return "fg"
    if nf == 6 and w == 128 and dir_c == 3 and field.appr_channels == 0:
        return "bg"
    raise NotImplementedError("lab4d_amd: no kernel instantiation for NeRF(num_freq_xyz=%d, W=%d, dir channels=%d, appr_channels=%d)"
                              % (nf, w, dir_c, field.appr_channels))


def warp_kind(field):
    """The fg_motion create_warp built (nnutils/warping.py:35-48): "rigid" (IdentityWarp, the reference's default, or a field without a
    warp), "dense" (a bare DenseWarp), "skinning" (bob / skel-*), "composed" (comp_skel-*_dense)."""
    warp = getattr(field, "warp", None)
    if warp is None:
        return "rigid"
    name = type(warp).__name__
    if name == "IdentityWarp":
        return "rigid"
    if name == "DenseWarp":
        return "dense"
    if name == "ComposedWarp":
        return "composed"
    if name == "SkinningWarp":
        return "skinning"
    raise NotImplementedError("lab4d_amd: no kernel path for warp %s (NVPWarp is outside the hot path, SURVEY.md section 2)" % name)


def dense_net_of(cond_mlp):
    """The chain-kernel instantiation of a DenseWarp map: D=2 (ComposedWarp's post-warp, warping.py:432-434) or the class default D=6 with the
    skip connection at layer 4 (fg_motion "dense", warping.py:94-141); W = 256 either way."""
    # read off the module's own layers (base.py:50-64: linear_1 .. linear_D, each a Sequential(Linear, ReLU), then linear_final)
    hidden = [getattr(cond_mlp, "linear_%d" % (i + 1)) for i in range(64) if hasattr(cond_mlp, "linear_%d" % (i + 1))]
    lin = [next(iter(h.parameters())) for h in hidden]  # the Linear's weight (out, in)
    D, Wd = len(lin), (lin[0].shape[0] if lin else 0)
    skips = [i for i in range(1, D) if lin[i].shape[1] != Wd]
    if Wd == 256 and D == 2:
        return mlp.NET_DENSE
    if Wd == 256 and D == 6 and skips == [4]:
        return mlp.NET_DENSE6
    raise NotImplementedError("lab4d_amd: no kernel instantiation for DenseWarp(D=%d, W=%d, skips=%s)" % (D, Wd, skips))


# ---------------------------------------------------------------------------------------------------
# per-module forwards
# ---------------------------------------------------------------------------------------------------
def nerf_forward(self, xyz, dir=None, frame_id=None, inst_id=None, get_density=True):
    """NeRF.forward (nnutils/nerf.py:167-215).  xyz (M,...,3); frame_id / inst_id (M,) or None."""
    if frame_id is not None:
        assert frame_id.ndim == 1
    if inst_id is not None:
        assert inst_id.ndim == 1
    P = field_params(self)
    M, dev = xyz.shape[0], xyz.device
    alpha = self.pos_embedding.alpha
    if field_kind(self) == "fg":
        fr = {"code_base": inst_code(self.basefield, inst_id, M, dev)}
        if dir is None:
            return DF.nerf_forward(P, xyz, fr, _prec(self), with_color=False, get_density=get_density, alpha=alpha)
        fr["code_color"] = inst_code(self.colorfield, inst_id, M, dev)
        fr["appr_code"] = self.appr_embedding.get_vals(frame_id)
        return DF.nerf_forward(P, xyz, fr, _prec(self), with_color=True, get_density=get_density, alpha=alpha)
    codes = {"basefield": inst_code(self.basefield, inst_id, M, dev)}
    if dir is not None:
        codes["colorfield"] = inst_code(self.colorfield, inst_id, M, dev)
    return DF.nerf_forward_bg(P, xyz, dir, codes, _prec(self), get_density=get_density, alpha=alpha)


def vis_forward(self, xyz, inst_id=None):
    """VisField.forward (nnutils/visibility.py:53-63)."""
    P = params_of(self, "vis_mlp.")
    fr = {"code_vis": inst_code(self.basefield, inst_id, xyz.shape[0], xyz.device)}
    return DF.vis_field(P, xyz, fr, _prec(self))


def compute_feat(self, xyz):
    """FeatureNeRF.compute_feat (nnutils/feature.py:136-150); train-only like the reference's decorator (decorator.py:4-17)."""
    if not self.training:
        return {}
    return {"feature": DF.compute_feat(field_params(self), xyz, _prec(self))}


def _articulations(self, frame_id, samples_dict):
    if "rest_articulation" in samples_dict and "t_articulation" in samples_dict:
        return samples_dict["t_articulation"], samples_dict["rest_articulation"]
    return self.articulation.get_vals_and_mean(frame_id)


def _skin_inputs(self, xyz, frame_id, inst_id, backward):
    """Time embedding and instance code of the delta-skin field (skinning.py:107-117): the forward warp runs at the mean
    time embedding (warping.py:314: frame_id = None)."""
    sk = self.skinning_model
    M = xyz.shape[0]
    if backward and frame_id is not None:
        te = sk.time_embedding(frame_id)
    else:
        te = sk.time_embedding.get_mean_embedding(xyz.device)
    return te, inst_code(sk.delta_field, inst_id, M, xyz.device)


def _warp_params(warp):
    P = {"warp." + k: v for k, v in params_of(warp).items()}
    sk = getattr(warp, "skinning_model", None)
    if sk is not None and getattr(sk, "symm_idx", None) is not None:
        P["warp.skinning_model.symm_idx"] = torch.as_tensor(sk.symm_idx, dtype=torch.long, device=P["warp.logibeta"].device)
    return P


def skinning_forward(self, xyz, frame_id, inst_id, backward=False, samples_dict={}, return_aux=False):
    """SkinningWarp.forward (nnutils/warping.py:277-336)."""
    t_art, rest_art = _articulations(self, frame_id, samples_dict)
    te, code = _skin_inputs(self, xyz, frame_id, inst_id, backward)
    out, aux = W.skinning_warp(_warp_params(self), xyz, t_art, rest_art, te, code, backward, _prec(self))
    return (out, aux) if return_aux else out


def dense_forward(self, xyz, frame_id, inst_id, backward=False, samples_dict={}, return_aux=False):
    """DenseWarp.forward (nnutils/warping.py:143-170)."""
    which = self.backward_map if backward else self.forward_map
    P = {"d." + k: v for k, v in params_of(self).items()}
    out = W.dense_warp(P, xyz, self.time_embedding(frame_id), inst_code(which, inst_id, xyz.shape[0], xyz.device), backward, _prec(self),
                       prefix="d.", net=dense_net_of(which))
    return (out, {}) if return_aux else out


def _dense_inputs(self, like, frame_id, inst_id):
    """Per-frame inputs of ComposedWarp's post-warp; `like`: any tensor with the frames on axis 0 (for M and the device)."""
    if frame_id is None:  # the post-warp is skipped (warping.py:460,474)
        return None
    pw, M, dev = self.post_warp, like.shape[0], like.device
    return {"t_embed": pw.time_embedding(frame_id), "code_fw": inst_code(pw.forward_map, inst_id, M, dev),
            "code_bw": inst_code(pw.backward_map, inst_id, M, dev)}


def composed_forward(self, xyz, frame_id, inst_id, backward=False, samples_dict={}, return_aux=False):
    """ComposedWarp.forward (nnutils/warping.py:445-483): skeleton skinning composed with the dense post-warp."""
    t_art, rest_art = _articulations(self, frame_id, samples_dict)
    te, code = _skin_inputs(self, xyz, frame_id, inst_id, backward)
    out, aux = W.composed_warp(_warp_params(self), xyz, t_art, rest_art, te, code, backward, _prec(self),
                               dense=_dense_inputs(self, xyz, frame_id, inst_id))
    return (out, aux) if return_aux else out


def appearance_get_vals(self, frame_id=None):
    """AppearanceEmbedding.get_vals (TimeMLP.get_vals, nnutils/time.py:107-117, with the AppearanceEmbedding.forward of
    appearance.py:46-56): Fourier(t) -> TimeEmbedding -> TimeMLP(D=2, W=64) -> Linear(64, 32).  M rows: rowmlp programs on the GPU since
    round 6 ("negligible when per-frame"); what matters for the per-sample path is that the result is consumed
    as a per-FRAME bias of the rgb head (lab4d_amd.mlp.pf_bias_of) instead of being broadcast to every sample
    (nerf.py:201-205) -- including in the compacted eval path, where the reference evaluates this MLP per SAMPLE
    (nerf.py:795-798)."""
    from . import pose
    return pose.appearance_vals(params_of(self, "appr."), "appr", frame_id, _time_info(self.time_embedding))


# ---------------------------------------------------------------------------------------------------
# field-level entry points
# ---------------------------------------------------------------------------------------------------
def _frames(self, samples_dict, need_color=True):
    """Per-frame inputs of the functional field code from the reference's samples_dict (deformable.py:254-289,
    nerf.py:530-578) + the per-frame modules of `self`."""
    frame_id, inst_id = samples_dict["frame_id"], samples_dict["inst_id"]
    M, dev = samples_dict["Kinv"].shape[0], samples_dict["Kinv"].device
    fr = {k: samples_dict[k] for k in ("Kinv", "field2cam", "near_far", "frame_id", "inst_id")}
    fr["code_base"] = inst_code(self.basefield, inst_id, M, dev)
    fr["code_color"] = inst_code(self.colorfield, inst_id, M, dev)
    fr["code_vis"] = inst_code(self.vis_mlp.basefield, inst_id, M, dev)
    if self.appr_channels > 0:
        fr["appr_code"] = self.appr_embedding.get_vals(frame_id)
    kind = warp_kind(self)
    if kind == "rigid":
        fr["motion"] = "rigid"
    elif kind == "dense":
        warp = self.warp
        dense_net_of(warp.forward_map)  # D=6, W=256 or a loud refusal
        fr["motion"] = "dense"
        fr["t_embed_dense"] = warp.time_embedding(frame_id)
        fr["code_dense_fw"] = inst_code(warp.forward_map, inst_id, M, dev)
        fr["code_dense_bw"] = inst_code(warp.backward_map, inst_id, M, dev)
    else:
        warp = self.warp
        fr["t_articulation"], fr["rest_articulation"] = _articulations(warp, frame_id, samples_dict)
        if "rest_articulation" in samples_dict and M % 2 == 0:
            # a caller-supplied rest articulation may differ between pair partners (the reference then warps into the partner's frame with
            # the PARTNER's rest, nerf.py:966-973): the shared-skinning-field shortcut holds only when it equals its flip_pair
            rest = fr["rest_articulation"]
            fr["rest_shared_in_pair"] = all(bool(torch.equal(r, r.view(M // 2, 2, *r.shape[1:]).flip(1).reshape(r.shape))) for r in rest)
        sk = warp.skinning_model
        fr["t_embed"] = sk.time_embedding(frame_id)
        fr["t_embed_mean"] = sk.time_embedding.get_mean_embedding(dev)
        fr["code_skin"] = inst_code(sk.delta_field, inst_id, M, dev)
        if kind == "composed":
            fr["dense"] = _dense_inputs(warp, samples_dict["Kinv"], frame_id, inst_id)
    if "feature" in samples_dict:
        fr["feature"] = samples_dict["feature"]
    return fr


def draw_rng(M, N, D, device):
    """The two host-side random draws of a training query, in the reference's order: the eikonal ray subset
    (torch.multinomial on a CPU tensor, nerf.py:437-440) then the matching candidates (torch.randperm on the CPU,
    feature.py:177)."""
    n = max(M * N // 16, 1)
    eik = torch.multinomial(torch.ones(M * N), n, replacement=False).to(device) if M * N > n else None
    S = M * N * D
    return {"eik_inds": eik, "eik_inds_bg": eik, "match_perm": torch.randperm(S)[: min(1024, S)].to(device)}


def query_field(self, samples_dict, flow_thresh=None):
    """NeRF.query_field (nnutils/nerf.py:580-684) including the FeatureNeRF (feature.py:89-134) and Deformable
    (deformable.py:300-327) overrides: (feat_dict, deltas, aux_dict) with the reference's keys."""
    P = field_params(self)
    fr = _frames(self, samples_dict)
    hxy = samples_dict["hxy"]
    alpha = self.pos_embedding.alpha
    kind = field_kind(self)
    if kind == "bg":
        if self.training:
            M, N = hxy.shape[:2]
            return DF.query_field_train_bg(P, fr, hxy, draw_rng(M, N, _ndepth(self), hxy.device), flow_thresh, _ndepth(self), alpha, _prec(self))
        fd, deltas, _ = DF.query_field_eval_bg(P, fr, hxy, _ndepth(self), alpha, _prec(self))
        return fd, deltas, {}
    if self.training:
        M, N = hxy.shape[:2]
        return DF.query_field_train(P, fr, hxy, draw_rng(M, N, _ndepth(self), hxy.device), flow_thresh, _ndepth(self), alpha, _prec(self))
    fd, deltas, _ = DF.query_field_eval(P, fr, hxy, _ndepth(self), alpha, _prec(self))
    return fd, deltas, {}


def backward_warp(self, xyz_cam, dir_cam, field2cam, frame_id, inst_id, samples_dict={}):
    """NeRF.backward_warp / Deformable.backward_warp (nnutils/nerf.py:865-887, deformable.py:119-152)."""
    from . import quat_utils as Q
    qi, ti = Q.quaternion_translation_inverse(field2cam[0], field2cam[1])
    xyz_t = DF.rigid_apply(qi, ti, xyz_cam)
    dir_f = DF.rigid_apply(qi, torch.zeros_like(ti), dir_cam)
    if getattr(self, "warp", None) is None:
        return {"xyz": xyz_t, "dir": dir_f, "xyz_t": xyz_t}
    xyz, aux = self.warp(xyz_t, frame_id, inst_id, backward=True, samples_dict=samples_dict, return_aux=True)
    out = {"xyz": xyz, "dir": dir_f, "xyz_t": xyz_t}
    out.update(aux)
    return out


def forward_warp(self, xyz, field2cam, frame_id, inst_id, samples_dict={}):
    """NeRF.forward_warp / Deformable.forward_warp (nnutils/nerf.py:889-903, deformable.py:154-171)."""
    if getattr(self, "warp", None) is not None:
        xyz = self.warp(xyz, frame_id, inst_id, samples_dict=samples_dict)
    return DF.rigid_apply(field2cam[0], field2cam[1], xyz)


def compose_fields(multifields_dict, deltas_dict):
    """MultiFields.compose_fields (nnutils/multifields.py:339-398)."""
    return multifields.compose_fields(multifields_dict, deltas_dict)


# ---------------------------------------------------------------------------------------------------
# dvr_model (engine/model.py)
# ---------------------------------------------------------------------------------------------------
def dvr_render_samples(self, samples_dict, flow_thresh=None):
    """dvr_model.render_samples (engine/model.py:328-361)."""
    fields, deltas, aux = {}, {}, {}
    for cat, field in self.fields.field_params.items():
        fields[cat], deltas[cat], aux[cat] = query_field(field, samples_dict[cat], flow_thresh=flow_thresh)
    comp, d = compose_fields(fields, deltas)
    rendered = dict(RU.render_pixel(comp, d))
    for cat in fields:
        aux[cat] = dict(aux[cat])
        aux[cat].update(RU.render_pixel(fields[cat], deltas[cat]))
    if "fg" in aux and "xyz_matches" in aux["fg"]:
        rendered["xyz_matches"] = aux["fg"]["xyz_matches"]
        rendered["xyz_reproj"] = aux["fg"]["xyz_reproj"]
    return {"rendered": rendered, "aux_dict": aux}


def dvr_render_samples_chunk(self, samples_dict, flow_thresh=None, chunk_size=8192):
    """dvr_model.render_samples_chunk (engine/model.py:259-326): chunks of ceil(chunk_size // M) pixels along N."""
    cat0 = list(samples_dict.keys())[0]
    M, N = samples_dict[cat0]["hxy"].shape[:2]
    num_chunks = int(math.ceil(M * N / chunk_size))
    chunk_n = int(math.ceil(chunk_size // M))
    rendered, aux = defaultdict(list), defaultdict(lambda: defaultdict(list))
    for i in range(num_chunks):
        # like the reference, only "hxy" is cut (model.py:299-305); "feature" follows it here because the matching loss needs
        # pixel-aligned features (the reference's training batches never exceed one chunk)
        sd = {}
        for cat, d in samples_dict.items():
            sd[cat] = dict(d)
            sd[cat]["hxy"] = d["hxy"][:, i * chunk_n:(i + 1) * chunk_n]
            if torch.is_tensor(d.get("feature")) and d["feature"].shape[1] == N:
                sd[cat]["feature"] = d["feature"][:, i * chunk_n:(i + 1) * chunk_n]
        if sd[cat0]["hxy"].shape[1] == 0:
            continue
        res = self.render_samples(sd, flow_thresh=flow_thresh)
        for k, v in res["rendered"].items():
            rendered[k].append(v)
        for cat, d in res["aux_dict"].items():
            for k, v in d.items():
                aux[cat][k].append(v)
    return {"rendered": {k: torch.cat(v, 1) for k, v in rendered.items()},
            "aux_dict": {c: {k: torch.cat(v, 1) for k, v in d.items()} for c, d in aux.items()}}


def dvr_render(self, batch, flow_thresh=None):
    """dvr_model.render (engine/model.py:217-235)."""
    samples_dict = self.get_samples(batch)
    return self.render_samples_chunk(samples_dict, flow_thresh=flow_thresh)


def dvr_evaluate(self, batch, is_pair=True):
    """dvr_model.evaluate (engine/model.py:162-209): frame (pair) by frame (pair), square images, masked by "mask"."""
    div = 2 if is_pair else 1
    self.process_frameid(batch)
    out = defaultdict(list)
    for i in range(len(batch["frameid"]) // div):
        sub = {}
        for k, v in batch.items():
            sub[k] = {k2: v2[i * div:(i + 1) * div] for k2, v2 in v.items()} if isinstance(v, dict) else v[i * div:(i + 1) * div]
        for k, v in self.render(sub)["rendered"].items():
            res = int(round(math.sqrt(v.shape[1])))
            out[k].append(v.view(div, res, res, -1)[0])
    out = {k: torch.stack(v, 0) for k, v in out.items()}
    for k in out:
        if "mask" not in k:
            out[k] = out[k] * out["mask"]
    return out


# ---------------------------------------------------------------------------------------------------
# loss epilogue and optimizer step
# ---------------------------------------------------------------------------------------------------
REG_FIELD_TERMS = ("reg_visibility", "reg_soft_deform", "reg_gauss_skin", "reg_cam_prior", "reg_skel_prior")
LOSS_ORDER = ("mask", "feature", "feat_reproj", "rgb", "depth", "flow", "vis", "reg_gauss_mask", "reg_visibility", "reg_eikonal", "reg_deform_cyc",
              "reg_delta_skin", "reg_skin_entropy", "reg_soft_deform", "reg_gauss_skin", "reg_cam_prior", "reg_skel_prior")


def dvr_compute_loss(self, batch, results):
    """dvr_model.compute_loss (engine/model.py:375-399) = compute_recon_loss + mask_losses + compute_reg_loss + apply_loss_weights.
    Every per-ray term (the eight reconstruction terms and the four rendered regularisers) comes out of ONE kernel pass each way
    (csrc/losses.hip through deformable.losses_fg / losses_comp: masked sums, positive counts, weights); the field-level regularisers
    (visibility decay, soft deformation, gaussian-skin consistency, camera / skeleton priors) are the reference's own small queries --
    their field evaluations already run on the kernels through the patched module forwards -- weighted by the reference's own
    apply_loss_weights.  Same keys, same values, same order as the reference's loss_dict."""
    config = self.config
    field_type = config["field_type"]
    if field_type == "fg":
        per_ray = DF.losses_fg(results, batch, config["train_res"], config)
    elif field_type == "comp":
        per_ray = DF.losses_comp(results, batch, config["train_res"], config)
    else:  # field_type "bg": its render runs on the kernels (query_field_train_bg); the loss epilogue stays the reference's own
        return _original("lab4d.engine.model.dvr_model.compute_loss")(self, batch, results)
    reg = {}
    self.compute_reg_loss(reg, results)  # the reference's method; the rendered terms it also lists are dropped, the kernel formed them
    reg = {k: v for k, v in reg.items() if k in REG_FIELD_TERMS}
    self.apply_loss_weights(reg, config)
    merged = dict(per_ray)
    merged.update(reg)
    return {k: merged[k] for k in LOSS_ORDER if k in merged}


def _original(name):
    for n, _, _, orig in _ORIGINALS:
        if n == name:
            return orig
    raise RuntimeError("lab4d_amd.patch: %s is not bound (call patch() first)" % name)


def trainer_optimizer_init(self, is_resumed=False):
    """Trainer.optimizer_init (engine/trainer.py:150-210): the reference's own selection of parameters and learning rates and its
    OneCycleLR stay as they are; the torch.optim.AdamW they were built around becomes a TorchFlatAdamW IN PLACE (one flat parameter /
    gradient / moment buffer, three launches per step, gradients accumulated by the weight-gradient kernels), so the scheduler, the
    two-rounds-back state cache and the checkpoints keep talking to the same object."""
    from . import mlp as _mlp
    from .optim import TorchFlatAdamW
    _original("lab4d.engine.trainer.Trainer.optimizer_init")(self, is_resumed)
    TorchFlatAdamW.adopt(self.optimizer)
    _mlp.FUSED_GRAD_ACCUM = True
    ddp_local_accumulation(getattr(self, "model", None))


def ddp_world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def ddp_local_accumulation(model):
    """Data-parallel training through the reference's own Trainer (engine/trainer.py:108-113 wraps the model in DistributedDataParallel with
    find_unused_parameters=False; lab4d/train.py:28-33 starts one process per GPU).  With FUSED_GRAD_ACCUM the weight-gradient kernels ADD into
    `weight.grad` -- views of the optimizer's flat fp32 buffer -- and hand autograd None for those inputs, so their AccumulateGrad hooks never fire and
    DDP's reducer would wait for them ("Expected to have finished reduction in the prior iteration").  The reduction therefore is what SURVEY 8e asks
    for anyway: ONE all-reduce (mean) of the flat gradient buffer per step, issued by trainer_check_grad in front of the clip (allreduce_flat_grad
    below) -- and DDP, which has already broadcast rank 0's weights at construction and keeps broadcasting the buffers, is told to stop reducing:
    require_backward_grad_sync = False is exactly what its no_sync() context sets.  Returns True when a DDP wrapper was switched."""
    from torch.nn.parallel import DistributedDataParallel
    if isinstance(model, DistributedDataParallel):
        model.require_backward_grad_sync = False
        return True
    return False


def ddp_keep_buffer_sync(model):
    """DDP re-broadcasts rank 0's buffers (aabb, near / far, proxy tables ...) at every forward only while its gradient reduction is on (its
    _post_forward clears `require_forward_param_sync` otherwise); with the reduction moved to allreduce_flat_grad the flag is set again once per
    iteration, so the reference's behaviour -- buffers follow rank 0 -- is kept."""
    from torch.nn.parallel import DistributedDataParallel
    if isinstance(model, DistributedDataParallel):
        model.require_forward_param_sync = True


def allreduce_flat_grad(opt):
    """The data-parallel collective of the patched Trainer: one all-reduce of TorchFlatAdamW's flat gradient buffer, averaged over the ranks (DDP's
    semantics: every rank normalises its loss over its own rays, gradients are averaged; RCCL over xGMI on the GPUs, gloo in the CPU tests).  No-op
    on one rank."""
    world = ddp_world()
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(opt.flat.flat_grad)
        opt.flat.flat_grad.div_(world)
    return world


def trainer_check_grad(self, thresh=5.0):
    """Trainer.check_grad (engine/trainer.py:581-604): clip to `thresh`; a step whose pre-clip norm exceeds it is discarded (the
    reference zeroes the gradients, after which torch's optimizer skips every parameter) and the state cached two rounds ago is loaded when
    there is one.  Norm, clip coefficient and the discard decision are formed on the device and applied by the following
    optimizer.step(); the host only looks at the decision when there is a cache to roll back to -- the one place the reference
    synchronises as well (`if grad_norm > thresh`)."""
    opt = self.optimizer
    allreduce_flat_grad(opt)  # (data-parallel runs: the rank-mean of the flat gradient, see ddp_local_accumulation)
    ddp_keep_buffer_sync(getattr(self, "model", None))
    grad_norm = opt.check_grad(thresh)
    if self.model_cache[0] is not None and int(opt.skipped):
        opt.zero_grad()
        print("large grad: %.2f, resume from cached weights" % float(grad_norm))
        self.model.load_state_dict(self.model_cache[0])
        self.optimizer.load_state_dict(self.optimizer_cache[0])
        self.scheduler.load_state_dict(self.scheduler_cache[0])


# ---------------------------------------------------------------------------------------------------
# proxy-geometry refresh and batch ingestion (row 4)
# ---------------------------------------------------------------------------------------------------
def nerf_extract_canonical_mesh(self, grid_size=64, level=0.0, inst_id=None, use_visibility=True, use_extend_aabb=True):
    """NeRF.extract_canonical_mesh (nnutils/nerf.py:303-343).  The dense grid query -- grid_size^3 evaluations of the sdf head and of the
    visibility field, the per-sample work of the proxy refresh -- is ONE inference-mode launch per network over the whole grid
    (lab4d_amd.proxy.grid_query: 0.7 ms at 64^3); iso-surface extraction, the unit-cube -> box transform and the connected-component filter
    stay the reference's own `geom_utils.marching_cubes` (CPU skimage / trimesh), which is handed the two finished volumes through the
    sdf_func / visibility_func it asks for chunk by chunk."""
    import importlib
    from . import proxy
    geom = importlib.import_module("lab4d.utils.geom_utils")
    P = field_params(self)
    dev = P["logibeta"].device if "logibeta" in P else next(self.parameters()).device
    iid = None if inst_id is None else torch.tensor([inst_id], device=dev)
    code_base = inst_code(self.basefield, iid, 1, dev)
    code_vis = inst_code(self.vis_mlp.basefield, iid, 1, dev)
    alpha = getattr(self.pos_embedding, "alpha", None)
    with torch.no_grad():
        # (round 5, ADVICE r04: the bg field is a NeRF too -- multifields.py:86-93 -- and MultiFields.update_geometry_aux meshes every field)
        sdf, vis, box = proxy.grid_query(P, self.aabb, grid_size=grid_size, code_base=code_base, code_vis=code_vis, prec=_prec(self),
                                         use_visibility=use_visibility, extend=0.5 if use_extend_aabb else 0.0, alpha=alpha, kind=field_kind(self))
    sdf, vis = sdf.reshape(-1, 1), vis.reshape(-1, 1)

    def served(vol):  # the volumes are final: marching_cubes' eval_func_chunk walks the grid in order, each call takes the next rows
        pos = [0]

        def f(xyz):
            out = vol[pos[0]:pos[0] + xyz.shape[0]]
            pos[0] += xyz.shape[0]
            return out
        return f
    return geom.marching_cubes(served(sdf), box, visibility_func=served(vis) if use_visibility else None, grid_size=grid_size, level=level,
                               apply_connected_component=True if self.category == "fg" else False)


def nerf_update_aabb(self, beta=0.9):
    """NeRF.update_aabb (nerf.py:345-356): the bounds of the new proxy mesh blended into the field's box, on the device."""
    from . import proxy
    bounds = self.proxy_geometry.bounds
    if bounds is not None:
        self.aabb = proxy.update_aabb(self.aabb, torch.as_tensor(bounds, dtype=torch.float32, device=self.aabb.device), beta)


def nerf_update_near_far(self, beta=0.9):
    """NeRF.update_near_far (nerf.py:358-376): depth range of the proxy vertices in every camera (quaternion / translation form straight from
    CameraMLP.get_vals, no 4x4 matrices), blended into the per-frame near / far table."""
    from . import proxy
    verts = self.proxy_geometry.vertices
    if verts is None:
        return
    dev = next(self.parameters()).device
    with torch.no_grad():
        quat, trans = self.camera_mlp.get_vals()
        pts = torch.as_tensor(verts, dtype=torch.float32, device=dev)
        fm = self.camera_mlp.time_embedding.frame_mapping
        self.near_far.data.copy_(proxy.update_near_far(self.near_far.data, fm, pts, quat, trans, beta))


def device_loader_of(ds, device="cuda"):
    """A lab4d_amd.ingest.DeviceVidLoader over one reference VidDataset (dataloader/vidloader.py:46-161): the arrays it memory-maps are
    uploaded once (FrameCache), after which a batch of frame pairs is one gather launch instead of per-pixel numpy reads in worker
    processes + collate + a host-to-device copy.  Cached on the dataset object."""
    from . import ingest
    dl = getattr(ds, "_lab4d_device_loader", None)
    if dl is None:
        mm = ds.mmap_list
        cache = ingest.FrameCache(mm["rgb"], mm["mask"], mm["depth"], mm["flowfw"], mm["flowbw"], mm["feature"], ds.crop2raw, ds.is_detected,
                                  dataid=ds.dataid, frame_map=list(ds.frame_info.frame_map), device=device)
        dl = ingest.DeviceVidLoader(cache, ds.delta_list, ds.pixels_per_image, load_pair=ds.load_pair)
        ds._lab4d_device_loader = dl
    return dl


def vid_load_data(self, im0idx):
    """VidDataset.load_data (vidloader.py:198-215) served from the HBM-resident frame cache: same keys, shapes and dtypes, DEVICE tensors
    (torch.utils.data's default collate stacks them as they are).  The pair's delta is drawn with the reference's own sample_delta (numpy
    RNG, like the reference); the pixels by the device-side sampler.  Opt-in (patch(ingest=True)): it needs the dataset in the process that
    owns the GPU, i.e. DataLoader(num_workers=0) -- the reference's worker processes never touch the device."""
    if self.pixels_per_image == -1:
        return _original("lab4d.dataloader.vidloader.VidDataset.load_data")(self, im0idx)  # full-frame reads (evaluation): the reference's path
    dl = device_loader_of(self)
    return dl.load_data(im0idx, delta=int(self.sample_delta(im0idx)))


INGEST_BINDINGS = [("lab4d.dataloader.vidloader", "VidDataset", "load_data", vid_load_data, False)]

# ---------------------------------------------------------------------------------------------------
# per-frame articulation: the kinematic tree behind every SkinningWarp call
# ---------------------------------------------------------------------------------------------------
def articulation_skel_forward(self, t_embed, inst_id, return_so3=False, override_so3=None, override_log_bone_len=None,
                              override_local_rest_joints=None):
    """ArticulationSkelMLP.forward (nnutils/pose.py:417-470).  The time MLP and the so3 head are one rowmlp program (round 6); the
    bone lengths, the forward kinematics over the tree (`fk_se3`: a Python loop of clone + matmul + index_put per joint in the
    reference), `matrix_to_quaternion` and `shift_joints_to_bones_dq` are ONE launch each way (csrc/fk.hip)."""
    from . import pose
    P = params_of(self, "a.")
    if override_so3 is None:
        lead = t_embed.shape[:-1]
        so3 = pose.articulation_so3(P, "a", t_embed.reshape(-1, t_embed.shape[-1])).reshape(*lead, self.num_se3, 3)
    else:
        so3 = override_so3
    if return_so3:
        return so3
    lead = so3.shape[:-2]
    so3_rows = so3.reshape(-1, self.num_se3, 3)
    if override_local_rest_joints is not None:  # reanimation: the caller supplies the parent-to-child offsets (pose.py:455-456)
        local = override_local_rest_joints.expand(*lead, self.num_se3, 3).reshape(-1, self.num_se3, 3)
        qr, qd = pose.fk_bones(local.contiguous(), so3_rows, self.edges, shift=P["a.shift"])
    else:
        if override_log_bone_len is not None:
            ll = override_log_bone_len.reshape(-1, self.num_se3)
        else:
            ii = None if inst_id is None else inst_id.reshape(-1)
            ll = pose.log_bone_len(P, "a.log_bone_len", ii, 1 if ii is None else ii.shape[0])
        skel = {"edges": self.edges, "symm_idx": self.symm_idx, "rest_joints": self.rest_joints}
        qr, qd = pose.skel_bones(so3_rows, ll, P["a.logscale"], skel, P["a.shift"])
    return qr.reshape(*lead, self.num_se3, 4), qd.reshape(*lead, self.num_se3, 4)


# ---------------------------------------------------------------------------------------------------
# per-frame MLPs: TimeEmbedding / CameraMLP / IntrinsicsMLP as rowmlp programs
# ---------------------------------------------------------------------------------------------------
def _time_info(te):
    """pose.time_info_of(te), cached on the module (the frame tables are fixed at construction; the cache follows the module across devices)."""
    from . import pose
    hit = te.__dict__.get("_lab4d_time_info")
    if hit is None or hit["frame_mapping"].device != te.frame_mapping.device:
        hit = pose.time_info_of(te)
        te.__dict__["_lab4d_time_info"] = hit
    return hit


def time_embedding_forward(self, frame_id=None):
    """TimeEmbedding.forward (nnutils/embedding.py:194-217): frame ids -> time coordinate -> Fourier features -> mapping1, the video's
    InstEmbedding row, mapping2 -- ONE launch (csrc/rowmlp.hip time prologue + two layers) instead of ~12."""
    from . import pose
    P = params_of(self, "te.")
    info = _time_info(self)
    if frame_id is None:
        return pose.time_embedding(P, "te", None, info)
    if not torch.is_tensor(frame_id):
        frame_id = torch.tensor(frame_id).to(self.frame_to_vid.device)
    if frame_id.ndim == 1:
        return pose.time_embedding(P, "te", frame_id, info)
    if frame_id.shape[-1] != 1:  # (the reference's PosEmbedding(1, F) takes the LAST axis as its single channel: anything else fails there too)
        raise RuntimeError("TimeEmbedding.forward: frame_id must be (M,) or (..., 1), got %s" % (tuple(frame_id.shape),))
    out = pose.time_embedding(P, "te", frame_id.reshape(-1), info)
    return out.reshape(*frame_id.shape[:-1], out.shape[-1])


def camera_get_vals(self, frame_id=None):
    """CameraMLP.get_vals (nnutils/pose.py:130-150): time prologue + TimeEmbedding + TimeMLP + both heads as ONE rowmlp program (12 layers, one
    launch forward, two backward), then normalize + the per-video base rotation through the library's quaternion_mul."""
    from . import pose
    return pose.camera_vals(params_of(self, "c."), "c", frame_id, _time_info(self.time_embedding))


def intrinsics_get_vals(self, frame_id=None):
    """IntrinsicsMLP.get_vals (nnutils/intrinsics.py:86-107): one rowmlp program (10 layers) + the per-video focal / principal-point algebra."""
    from . import pose
    return pose.intrinsics_vals(params_of(self, "k."), "k", frame_id, _time_info(self.time_embedding))


# ---------------------------------------------------------------------------------------------------
# binding
# ---------------------------------------------------------------------------------------------------
# (module path, class name or None for module-level, attribute, replacement, is_static)
def bindings():
    from . import quaternion as _q  # noqa: F401  (registers the dqtorch-compatible package)
    return [
        ("lab4d.utils.render_utils", None, "sample_cam_rays", RU.sample_cam_rays, False),
        ("lab4d.utils.render_utils", None, "render_pixel", RU.render_pixel, False),
        ("lab4d.utils.render_utils", None, "compute_weights", RU.compute_weights, False),
        ("lab4d.utils.render_utils", None, "integrate", RU.integrate, False),
        ("lab4d.utils.render_utils", None, "sample_pdf", RU.sample_pdf, False),
        ("lab4d.nnutils.nerf", "NeRF", "forward", nerf_forward, False),
        ("lab4d.nnutils.nerf", "NeRF", "query_field", query_field, False),
        ("lab4d.nnutils.nerf", "NeRF", "backward_warp", backward_warp, False),
        ("lab4d.nnutils.nerf", "NeRF", "forward_warp", forward_warp, False),
        ("lab4d.nnutils.feature", "FeatureNeRF", "query_field", query_field, False),
        ("lab4d.nnutils.feature", "FeatureNeRF", "compute_feat", compute_feat, False),
        ("lab4d.nnutils.deformable", "Deformable", "query_field", query_field, False),
        ("lab4d.nnutils.deformable", "Deformable", "backward_warp", backward_warp, False),
        ("lab4d.nnutils.deformable", "Deformable", "forward_warp", forward_warp, False),
        ("lab4d.nnutils.visibility", "VisField", "forward", vis_forward, False),
        ("lab4d.nnutils.warping", "SkinningWarp", "forward", skinning_forward, False),
        ("lab4d.nnutils.warping", "DenseWarp", "forward", dense_forward, False),
        ("lab4d.nnutils.warping", "ComposedWarp", "forward", composed_forward, False),
        ("lab4d.nnutils.multifields", "MultiFields", "compose_fields", compose_fields, True),
        ("lab4d.nnutils.appearance", "AppearanceEmbedding", "get_vals", appearance_get_vals, False),
        ("lab4d.nnutils.pose", "ArticulationSkelMLP", "forward", articulation_skel_forward, False),
        ("lab4d.nnutils.embedding", "TimeEmbedding", "forward", time_embedding_forward, False),
        ("lab4d.nnutils.pose", "CameraMLP", "get_vals", camera_get_vals, False),
        ("lab4d.nnutils.intrinsics", "IntrinsicsMLP", "get_vals", intrinsics_get_vals, False),
        ("lab4d.engine.model", "dvr_model", "render", dvr_render, False),
        ("lab4d.engine.model", "dvr_model", "evaluate", dvr_evaluate, False),
        ("lab4d.engine.model", "dvr_model", "render_samples", dvr_render_samples, False),
        ("lab4d.engine.model", "dvr_model", "render_samples_chunk", dvr_render_samples_chunk, False),
        ("lab4d.engine.model", "dvr_model", "compute_loss", dvr_compute_loss, False),
        ("lab4d.engine.trainer", "Trainer", "optimizer_init", trainer_optimizer_init, False),
        ("lab4d.engine.trainer", "Trainer", "check_grad", trainer_check_grad, False),
        ("lab4d.nnutils.nerf", "NeRF", "extract_canonical_mesh", nerf_extract_canonical_mesh, False),
        ("lab4d.nnutils.nerf", "NeRF", "update_aabb", nerf_update_aabb, False),
        ("lab4d.nnutils.nerf", "NeRF", "update_near_far", nerf_update_near_far, False),
    ]


# names other reference modules imported BY VALUE from render_utils (`from lab4d.utils.render_utils import ...`,
# nerf.py:31, engine/model.py:14): rebinding the source module alone would not reach them
BY_VALUE = [("lab4d.nnutils.nerf", "sample_cam_rays"), ("lab4d.nnutils.nerf", "sample_pdf"), ("lab4d.nnutils.nerf", "compute_weights"),
            ("lab4d.engine.model", "render_pixel")]

_ORIGINALS = []
_INHERITED = object()


def install_quaternion():
    """`from quaternion import ...` (utils/quat_transform.py:15-16) resolves to the HIP library from here on."""
    from . import quaternion
    sys.modules["quaternion"] = quaternion
    return quaternion


def patch(precision="bf16", n_depth=64, ingest=False):
    """Rebind the reference's operator API to the HIP library.  `lab4d` must be importable.  Idempotent; `unpatch()`
    restores the originals.  Returns the list of "module.Class.attr" names that were rebound.
    ingest=True additionally serves VidDataset.load_data from an HBM-resident frame cache (INGEST_BINDINGS; needs DataLoader(num_workers=0))."""
    import importlib
    global PRECISION, N_DEPTH
    PRECISION = _PREC_NAMES[precision]
    N_DEPTH = int(n_depth)
    if _ORIGINALS:
        return [n for n, *_ in _ORIGINALS]
    install_quaternion()
    done = []
    for modname, cls, attr, fn, static in bindings() + (INGEST_BINDINGS if ingest else []):
        mod = importlib.import_module(modname)
        owner = mod if cls is None else getattr(mod, cls)
        # a method inherited from a base class (AppearanceEmbedding.get_vals lives on TimeMLP) is shadowed on the subclass and
        # the shadow removed again by unpatch(); everything else is swapped in place
        orig = owner.__dict__.get(attr, _INHERITED) if cls else getattr(owner, attr)
        _ORIGINALS.append(("%s.%s%s" % (modname, cls + "." if cls else "", attr), owner, attr, orig))
        setattr(owner, attr, staticmethod(fn) if static else fn)
        done.append(_ORIGINALS[-1][0])
    for modname, attr in BY_VALUE:
        mod = importlib.import_module(modname)
        if hasattr(mod, attr):
            _ORIGINALS.append(("%s.%s" % (modname, attr), mod, attr, getattr(mod, attr)))
            setattr(mod, attr, getattr(RU, attr))
    return done


def unpatch():
    while _ORIGINALS:
        _, owner, attr, orig = _ORIGINALS.pop()
        if orig is _INHERITED:
            delattr(owner, attr)
        else:
            setattr(owner, attr, orig)
