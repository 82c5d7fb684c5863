"""Host-side mirror of the reference's foreground field on the HIP kernels:
Deformable.query_field (lab4d/nnutils/deformable.py:300-327 -> feature.py:89-134 -> nerf.py:580-684),
NeRF.forward (nerf.py:167-215), backward/forward warps (deformable.py:119-171), compute_flow
(nerf.py:948-997), cycle_loss (deformable.py:173-198), compute_feat / global_match / forward_project
(feature.py:136-226), compute_gauss_density (deformable.py:329-356) and dvr_model.render_samples
(engine/model.py:328-361) for field_type == "fg".

Functional style: `P` maps the reference's state_dict names to device tensors, `fr` holds the per-frame
inputs produced by the per-frame modules (camera / articulation / appearance / time / instance codes).
Heavy per-sample work runs in liblab4d_hip.so; what is left in torch here is per-frame or per-ray glue plus
a handful of element-wise epilogues that DESIGN.md lists as not yet folded into kernels.
"""
import ctypes
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib, mlp
from . import quat_utils as Q
from . import render_utils as RU
from .warping import composed_warp, dense_warp, skinning_warp, skinning_warp_forward_multi

vp, ci, cf = _lib.vp, _lib.ci, _lib.cf
_lib.register("lab4d_gauss_density_forward", [vp, vp, ci, vp, ci, vp, vp, vp])
_lib.register("lab4d_gauss_density_backward", [vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, vp])
_lib.register("lab4d_l2_normalize_forward", [vp, ci, ci, vp, vp])
_lib.register("lab4d_l2_normalize_backward", [vp, vp, ci, ci, vp, vp])
_lib.register("lab4d_flow_cyc_forward", [vp] * 7 + [ctypes.c_long, ci, ci, cf, vp, vp, vp])
_lib.register("lab4d_flow_cyc_backward", [vp] * 8 + [ctypes.c_long, ci, ci, vp, vp, vp, vp, vp])
_lib.register("lab4d_volsdf_forward", [vp, vp, ctypes.c_long, vp, vp])
_lib.register("lab4d_volsdf_backward", [vp, vp, vp, ctypes.c_long, vp, vp, vp])


def flip_pair(x):
    """NeRF.flip_pair (nerf.py:929-946)."""
    if torch.is_tensor(x):
        if len(x) < 2:
            return x
        return x.view(x.shape[0] // 2, 2, -1).flip(1).view(x.shape)
    if isinstance(x, tuple):
        return tuple(flip_pair(t) for t in x)
    if isinstance(x, dict):
        return {k: flip_pair(v) for k, v in x.items()}
    return x


def _spf(x):
    n = 1
    for d in x.shape[1:-1]:
        n *= d
    return n


def rigid_apply(q, t, x):
    """quaternion_translation_apply with per-frame (M,4),(M,3) broadcast over (M,...,3) points
    (field_to_cam, nerf.py:846-863):  (w^2 - v.v) x + 2 (v.x) v + 2 w (v x x) + t."""
    shp = (x.shape[0],) + (1,) * (x.dim() - 2)
    w = q[:, 0].view(shp + (1,))
    v = q[:, 1:].view(shp + (3,))
    out = (w * w - (v * v).sum(-1, keepdim=True)) * x + 2 * (v * x).sum(-1, keepdim=True) * v + 2 * w * torch.linalg.cross(v.expand_as(x), x)
    return out + t.view(shp + (3,))


def pinhole_projection(Kmat, xyz_cam):
    """geom_utils.py:14-27."""
    shp = (xyz_cam.shape[0],) + (1,) * (xyz_cam.dim() - 2)
    K = Kmat.view(shp + (3, 3))
    hxy = (K * xyz_cam.unsqueeze(-2)).sum(-1)
    return hxy / (hxy[..., -1:] + 1e-6)


class GaussDensity(Function):
    @staticmethod
    def forward(ctx, xyz, centres, ibeta):
        xyz, centres = xyz.contiguous(), centres.contiguous()
        _lib.require_device(xyz, centres)
        S, B = xyz.shape[0], centres.shape[0]
        out = torch.empty(S, 1, device=xyz.device)
        best = torch.empty(S, dtype=torch.int32, device=xyz.device)
        ib = ibeta.detach().reshape(1).float().contiguous()  # stays on the device: no host sync (graph-capturable)
        _lib.check(_lib.lib().lab4d_gauss_density_forward(_lib.ptr(xyz), _lib.ptr(centres), B, _lib.ptr(ib), S, _lib.ptr(out), _lib.ptr(best),
                                                          _lib.stream()), "gauss_density_forward")
        ctx.save_for_backward(xyz, centres, best, ib)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xyz, centres, best, ib = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.empty_like(xyz)
        gc = torch.zeros_like(centres)
        gi = torch.zeros(1, device=xyz.device)
        _lib.check(_lib.lib().lab4d_gauss_density_backward(_lib.ptr(xyz), _lib.ptr(centres), centres.shape[0], _lib.ptr(ib), _lib.ptr(best), _lib.ptr(g),
                                                           xyz.shape[0], _lib.ptr(gx), _lib.ptr(gc), _lib.ptr(gi), _lib.stream()),
                   "gauss_density_backward")
        return gx, gc, gi


def gauss_density(P, xyz, rest_articulation, ft=None):
    """Deformable.compute_gauss_density (deformable.py:329-356): bones of frame 0 at rest."""
    if ft is not None:
        centre, ibeta = ft["gauss_centre"], ft["gauss_ibeta"]
    else:
        _, centre = Q.dual_quaternion_to_quaternion_translation((rest_articulation[0][:1], rest_articulation[1][:1]))
        ibeta = P["warp.logibeta"].exp()
    out = GaussDensity.apply(xyz.reshape(-1, 3), centre[0], ibeta)
    return out.view(xyz.shape[:-1] + (1,))


def _flip_yz(n):
    """normal * [1, -1, -1] (nerf.py:489-491) without building a constant on the host (hipGraph-capturable)."""
    return torch.cat([n[..., :1], -n[..., 1:]], -1)


class FlowCyc(Function):
    """compute_flow's projection into the pair partner's camera (nerf.py:948-997) and cycle_loss' distance
    (deformable.py:189-193) in one kernel each way (csrc/flow.hip).  xyz_next (M,N,D,3), q (M,4), t (M,3), Kmat (M,3,3): the
    partner's pose / intrinsics, hxy (M,N,3), xyz_cyc / xyz_t (M,N,D,3) or None -> flow (M,N,D,3) = (u, v, valid), cyc (M,N,D,1)."""

    @staticmethod
    def forward(ctx, xyz_next, q, t, Kmat, hxy, xyz_cyc, xyz_t, flow_thresh):
        xyz_next, q, t, Kmat, hxy = [x.contiguous().float() for x in (xyz_next, q, t, Kmat, hxy)]
        has_cyc = xyz_cyc is not None
        if has_cyc:
            xyz_cyc, xyz_t = xyz_cyc.contiguous().float(), xyz_t.contiguous().float()
        _lib.require_device(xyz_next, q, t, Kmat, hxy, xyz_cyc, xyz_t)
        M, N, D = xyz_next.shape[:3]
        S = M * N * D
        flow = torch.empty(M, N, D, 3, device=xyz_next.device)
        cyc = torch.empty(M, N, D, 1, device=xyz_next.device) if has_cyc else None
        _lib.check(_lib.lib().lab4d_flow_cyc_forward(_lib.ptr(xyz_next), _lib.ptr(q), _lib.ptr(t), _lib.ptr(Kmat), _lib.ptr(hxy), _lib.ptr(xyz_cyc),
                                                     _lib.ptr(xyz_t), S, N * D, D, -1.0 if flow_thresh is None else float(flow_thresh), _lib.ptr(flow),
                                                     _lib.ptr(cyc), _lib.stream()), "flow_cyc_forward")
        ctx.save_for_backward(xyz_next, q, t, Kmat, xyz_cyc, xyz_t)
        ctx.dims = (M, N, D)
        return (flow, cyc) if has_cyc else (flow, None)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_flow, g_cyc):
        xyz_next, q, t, Kmat, xyz_cyc, xyz_t = ctx.saved_tensors
        M, N, D = ctx.dims
        S = M * N * D
        has_cyc = xyz_cyc is not None
        g_flow = g_flow.contiguous()
        if has_cyc:
            g_cyc = torch.zeros(M, N, D, 1, device=g_flow.device) if g_cyc is None else g_cyc.contiguous()
        g_next = torch.empty_like(xyz_next)
        g_pf = torch.empty(M, 16, device=g_flow.device)
        g_xc = torch.empty_like(xyz_cyc) if has_cyc else None
        g_xt = torch.empty_like(xyz_t) if has_cyc else None
        _lib.check(_lib.lib().lab4d_flow_cyc_backward(_lib.ptr(xyz_next), _lib.ptr(q), _lib.ptr(t), _lib.ptr(Kmat), _lib.ptr(xyz_cyc), _lib.ptr(xyz_t),
                                                      _lib.ptr(g_flow), _lib.ptr(g_cyc) if has_cyc else None, S, N * D, M, _lib.ptr(g_next), _lib.ptr(g_pf),
                                                      _lib.ptr(g_xc), _lib.ptr(g_xt), _lib.stream()), "flow_cyc_backward")
        return g_next, g_pf[:, :4], g_pf[:, 4:7], g_pf[:, 7:].reshape(M, 3, 3), None, g_xc, g_xt, None


class VolSdfDensity(Function):
    """The VolSDF density of NeRF.forward (nerf.py:186-192) in one kernel each way (the reference: 8 element-wise launches)."""

    @staticmethod
    def forward(ctx, sdf, ibeta):
        sdf_c, ib = sdf.contiguous().float(), ibeta.reshape(1).contiguous().float()
        _lib.require_device(sdf_c, ib)
        out = torch.empty_like(sdf_c)
        _lib.check(_lib.lib().lab4d_volsdf_forward(_lib.ptr(sdf_c), _lib.ptr(ib), sdf_c.numel(), _lib.ptr(out), _lib.stream()), "volsdf_forward")
        ctx.save_for_backward(sdf_c, ib)
        ctx.ib_shape = ibeta.shape
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        sdf, ib = ctx.saved_tensors
        g = g.contiguous()
        g_sdf = torch.empty_like(sdf)
        g_ib = torch.empty(1, device=sdf.device) if ctx.needs_input_grad[1] else None
        _lib.check(_lib.lib().lab4d_volsdf_backward(_lib.ptr(sdf), _lib.ptr(ib), _lib.ptr(g), sdf.numel(), _lib.ptr(g_sdf), _lib.ptr(g_ib), _lib.stream()),
                   "volsdf_backward")
        return g_sdf, None if g_ib is None else g_ib.view(ctx.ib_shape)


def volsdf_density(sdf, logibeta):
    return VolSdfDensity.apply(sdf, logibeta.exp())


# The eikonal term's primal pass takes its ReLU pattern from the basefield pass over all samples (mlp.EikonalSdf); 0: it runs its own forward (A/B).
EIK_REUSE = os.environ.get("LAB4D_EIK_REUSE", "1") != "0"


def posenc_window(alpha, n_freq, device):
    """PosEmbedding.apply_annealing window (embedding.py:112-125)."""
    if alpha is None:
        return None
    w = torch.clamp(alpha * n_freq - torch.arange(n_freq, dtype=torch.float32, device=device), 0.0, 1.0)
    return 0.5 * (1 + torch.cos(torch.pi * w + torch.pi))


def nerf_forward(P, xyz, fr, prec, with_color=True, get_density=True, alpha=None, ft=None, tap=None):
    """NeRF.forward (nerf.py:167-215), fg configuration (no view dependence, appearance code in the rgb head).
    ft: the step's per-frame terms (frame_terms) -- the per-frame bias tables are taken from there instead of being re-formed."""
    shape = xyz.shape
    spf = _spf(xyz)
    x = xyz.reshape(-1, 3)
    dev = x.device
    # without the colour net nobody consumes layer 8: no export (inference mode then stores nothing at all; with only the points requiring a
    # gradient -- the eval path's normals -- the chain runs in its point-gradient-only mode: sign words + embedding, no activations, no dZ)
    r = mlp.run_chain(mlp.NET_FG_BASE, prec, P, x, spf, conds={0: fr["code_base"], 4: fr["code_base"]}, export_layer=8 if with_color else None,
                      freq_w=posenc_window(alpha, 10, dev), pfs_pre=None if ft is None else {0: ft["pf.base0"], 4: ft["pf.base4"]}, tap=tap)
    sdf, feat = r if with_color else (r, None)
    sdf = sdf.view(shape[:-1] + (1,))
    if get_density:
        out = volsdf_density(sdf, P["logibeta"])  # VolSDF (nerf.py:186-192)
    else:
        out = sdf
    if not with_color:
        return out
    rgb = mlp.run_chain(mlp.NET_FG_COLOR, prec, P, x, spf, conds={0: fr["code_color"], 3: fr["appr_code"]}, ext=feat,
                        freq_w=posenc_window(alpha, 12, dev), pfs_pre=None if ft is None else {0: ft["pf.color0"], 3: ft["pf.color3"]})
    return torch.sigmoid(rgb).view(shape[:-1] + (3,)), out


def nerf_forward_bg(P, xyz, dir, codes, prec, get_density=True, alpha=None, prefix="", tap=None):
    """NeRF.forward (nerf.py:167-215) for the background field NeRF(num_freq_xyz=6, num_freq_dir=0, appr_channels=0, D=5,
    W=128) (multifields.py:86-93): the rgb head sees [feature | raw view direction].  codes = {"basefield": (M,32),
    "colorfield": (M,32)}; dir (M,N,D,3) or None (sdf / density only)."""
    shape = xyz.shape
    spf = _spf(xyz)
    x = xyz.reshape(-1, 3)
    dev = x.device
    r = mlp.run_chain(mlp.NET_BG_BASE, prec, P, x, spf, conds={0: codes["basefield"], 4: codes["basefield"]}, export_layer=5 if dir is not None else None,
                      freq_w=posenc_window(alpha, 6, dev), prefix=prefix, tap=tap)  # (no export without the colour net: see nerf_forward)
    sdf, feat = r if dir is not None else (r, None)
    sdf = sdf.view(shape[:-1] + (1,))
    if get_density:
        out = volsdf_density(sdf, P[prefix + "logibeta"])
    else:
        out = sdf
    if dir is None:
        return out
    rgb = mlp.run_chain(mlp.NET_BG_COLOR, prec, P, x, spf, conds={0: codes["colorfield"]}, ext=feat, freq_w=posenc_window(alpha, 8, dev),
                        prefix=prefix, x2=dir.reshape(-1, 3))
    return torch.sigmoid(rgb).view(shape[:-1] + (3,)), out


def vis_field(P, xyz, fr, prec, ft=None):
    """VisField.forward (visibility.py:53-63)."""
    out = mlp.run_chain(mlp.NET_VIS, prec, P, xyz.reshape(-1, 3), _spf(xyz), conds={0: fr["code_vis"]},
                        pfs_pre=None if ft is None else {0: ft["pf.vis0"]})
    return out.view(xyz.shape[:-1] + (1,))


class L2Normalize(Function):
    """x / ||x||_2 over the last axis (feature.py:149-150) in one kernel each way."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        _lib.require_device(x)
        S, C = x.shape
        y = torch.empty_like(x)
        _lib.check(_lib.lib().lab4d_l2_normalize_forward(_lib.ptr(x), S, C, _lib.ptr(y), _lib.stream()), "l2_normalize_forward")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.empty_like(x)
        _lib.check(_lib.lib().lab4d_l2_normalize_backward(_lib.ptr(x), _lib.ptr(g), x.shape[0], x.shape[1], _lib.ptr(gx), _lib.stream()),
                   "l2_normalize_backward")
        return gx


def compute_feat(P, xyz, prec):
    """FeatureNeRF.compute_feat (feature.py:136-150)."""
    f = mlp.run_chain(mlp.NET_FEAT, prec, P, xyz.reshape(-1, 3), _spf(xyz))
    return L2Normalize.apply(f).view(xyz.shape[:-1] + (16,))


_lib.register("lab4d_global_match_workspace_floats", [ctypes.c_int])
_lib.register("lab4d_global_match_forward", [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3)
_lib.register("lab4d_global_match_backward", [ctypes.c_void_p] * 7 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 5)


class GlobalMatch(Function):
    """softmax((feat_px @ feat_c^T) * exp(logsigma), 1) @ xyz_c (feature.py:180-196) without the (R, K) score matrix: one kernel forward, the
    candidate-side adjoint + a partial-sum reduction backward (csrc/match.hip).  feat_px: the batch's pixel features (data, no gradient)."""

    @staticmethod
    def forward(ctx, feat_px, feat_c, xyz_c, logsigma):
        feat_px, feat_c, xyz_c, logsigma = [t.detach().contiguous().float() for t in (feat_px, feat_c, xyz_c, logsigma)]
        _lib.require_device(feat_px, feat_c, xyz_c, logsigma)
        R, C = feat_px.shape
        K = feat_c.shape[0]
        out = torch.empty(R, 3, device=feat_px.device)
        stats = torch.empty(R, 2, device=feat_px.device)
        with _lib.timed("k_match_fwd", (2.0 * R * K * (C + 3), 4.0 * (R * (C + 5) + K * (C + 3)))):
            _lib.check(_lib.lib().lab4d_global_match_forward(_lib.ptr(feat_px), _lib.ptr(feat_c), _lib.ptr(xyz_c), _lib.ptr(logsigma), R, C, K,
                                                             _lib.ptr(out), _lib.ptr(stats), _lib.stream()), "global_match_forward")
        ctx.save_for_backward(feat_px, feat_c, xyz_c, logsigma, out, stats)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("GlobalMatch: the pixel features are data (samples_dict['feature'], feature.py:127): no gradient path")
        feat_px, feat_c, xyz_c, logsigma, out, stats = ctx.saved_tensors
        R, C = feat_px.shape
        K = feat_c.shape[0]
        g = g.contiguous().float()
        g_fc, g_xc, g_ls = torch.empty_like(feat_c), torch.empty_like(xyz_c), torch.empty_like(logsigma)
        work = torch.empty(_lib.lib().lab4d_global_match_workspace_floats(K), device=g.device)
        with _lib.timed("k_match_bwd", (2.0 * R * K * (2 * C + 8), 4.0 * (R * (C + 8) + 2 * K * (C + 3)))):
            _lib.check(_lib.lib().lab4d_global_match_backward(_lib.ptr(feat_px), _lib.ptr(feat_c), _lib.ptr(xyz_c), _lib.ptr(logsigma), _lib.ptr(out),
                                                              _lib.ptr(stats), _lib.ptr(g), R, C, K, _lib.ptr(g_fc), _lib.ptr(g_xc), _lib.ptr(g_ls),
                                                              _lib.ptr(work), _lib.stream()), "global_match_backward")
        return None, g_fc, g_xc, g_ls


class RowTap(Function):
    """(x, x[idx]) for a dense per-sample tensor x (S, C) and <= 1024 drawn rows idx (the matching candidates, feature.py:176-178).  The first
    output is x itself: every other consumer reads THAT, so autograd hands backward() the summed dense gradient and the few gathered rows'
    gradient is added into it in place -- instead of a zero-filled (S, C) tensor, a scatter into it and one more dense add (3 GB of traffic per
    chunk for the (S,16) features)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.x_shape = x.shape
        return x.view_as(x), x.detach()[idx]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_dense, g_rows):
        (idx,) = ctx.saved_tensors
        if g_dense is None:  # nothing else consumed x
            g_dense = torch.zeros(ctx.x_shape, device=idx.device)
        elif not g_dense.is_contiguous():
            g_dense = g_dense.contiguous()
        elif g_rows is not None and (g_dense._use_count() > 2 or g_dense._is_view()):
            # autograd does not promise a private buffer: a producer can hand ONE tensor to several edges (AddBackward does), and a
            # caller-supplied gradient is the caller's.  Owned = this frame's argument + the engine's input buffer hold the only two
            # references and it is no view of something else; anything else is copied before the rows are added
            g_dense = g_dense.clone()
        if g_rows is not None:
            g_dense.index_add_(0, idx, g_rows.to(g_dense.dtype))  # in place on the gradient buffer autograd built for this edge
        return g_dense, None


def global_match(P, feat_px, feat_rows, xyz_rows):
    """FeatureNeRF.global_match (feature.py:152-199): soft arg-max over <= 1024 host-drawn candidates, given their gathered rows."""
    shape = feat_px.shape
    out = GlobalMatch.apply(feat_px.reshape(-1, shape[-1]), feat_rows, xyz_rows, P["logsigma"].reshape(1))
    return out.view(shape[:-1] + (3,))


def eikonal_subsample(P, xyz, code, rand_inds, alpha=None, prec=mlp.PREC_F32, net=mlp.NET_FG_BASE, prefix="", pf_tables=None, tap=None):
    """NeRF.compute_eikonal (nerf.py:416-453) on the host-drawn 1/16 ray subset: (|d sdf/dx| - 1)^2 with gradients to the
    basefield / sdf weights, via the primal + tangent-mode chain kernels (mlp.EikonalSdf) -- no second-order autograd.
    pf_tables = (pf0, pf4): the (M, mout) per-frame bias tables of layers 0 and 4 from the step's prologue; the per-ray rows are
    gathered from them (cond[idx] @ W^T == (cond @ W^T)[idx])."""
    M, N, D, _ = xyz.shape
    pts = xyz.reshape(M * N, D, 3)
    out = torch.zeros(M * N, D, device=xyz.device)
    if rand_inds is None:
        rand_inds = torch.arange(M * N, device=xyz.device)
    ray_frame = torch.div(rand_inds, N, rounding_mode="floor")
    x = pts[rand_inds].detach().reshape(-1, 3)
    if pf_tables is not None:
        ray_code, pf_rows = None, (pf_tables[0].index_select(0, ray_frame), pf_tables[1].index_select(0, ray_frame))
    else:
        ray_code, pf_rows = code[ray_frame], None
    eik_tap = None
    if tap and D % 64 == 0:  # the field was just evaluated on every sample: ray r = 64-sample blocks r D/64 .. of that pass (mlp.EikonalSdf)
        eik_tap = (tap, (rand_inds[:, None] * (D // 64) + torch.arange(D // 64, device=xyz.device)).reshape(-1))
    e = mlp.eikonal_sdf(P, x, ray_code, D, prec, freq_w=posenc_window(alpha, mlp.describe(net).n_freq, xyz.device), prefix=prefix, net=net,
                        pf_rows=pf_rows, tap=eik_tap)
    out = out.index_put((rand_inds,), e.view(-1, D))
    return out.reshape(M, N, D, 1)


def frame_terms(P, fr):
    """Everything the training-mode query derives from the weights and the per-frame inputs ALONE -- camera inverses, the
    bone transforms of the three skinning warps, the gaussian bone scales, the per-frame bias tables code @ W[:, cond]^T of
    every conditioned layer -- as a flat dict of tensors.  None of it depends on the rays, so a training step evaluates it ONCE
    (FramePrologue) instead of once per ray chunk; called inline (ft=None paths) it is the same arithmetic in the same order.
    SkinningWarp foreground only: with a dense post-warp (fr["dense"]) the warp terms stay inline."""
    from .warping import SKIN_AFFINE, get_gauss, skin_affine_table, skin_cond
    ft = {}
    M = fr["field2cam"][0].shape[0]
    ft["cam2field.q"], ft["cam2field.t"] = Q.quaternion_translation_inverse(fr["field2cam"][0], fr["field2cam"][1])
    nxt = flip_pair({k: fr.get(k) for k in ["Kinv", "field2cam", "t_articulation"]})
    ft["Kmat"], ft["Kmat_next"] = Q.kmatinv(fr["Kinv"]), Q.kmatinv(nxt["Kinv"])
    ft["field2cam_next.q"], ft["field2cam_next.t"] = nxt["field2cam"][0].contiguous(), nxt["field2cam"][1].contiguous()
    ft["scale"] = P["logscale"].exp()
    motion = fr.get("motion", "skinning")
    if motion not in ("rigid", "dense"):  # gaussian-bone density: SkinningWarp fields only (deformable.py:344)
        _, centre = Q.dual_quaternion_to_quaternion_translation((fr["rest_articulation"][0][:1], fr["rest_articulation"][1][:1]))
        ft["gauss_centre"], ft["gauss_ibeta"] = centre, P["warp.logibeta"].exp()
    W = lambda net, l: P[mlp.bindings(net)[l].wname]
    ft["pf.base0"] = mlp.pf_bias_of(mlp.NET_FG_BASE, 0, W(mlp.NET_FG_BASE, 0), fr["code_base"])
    ft["pf.base4"] = mlp.pf_bias_of(mlp.NET_FG_BASE, 4, W(mlp.NET_FG_BASE, 4), fr["code_base"])
    ft["pf.color0"] = mlp.pf_bias_of(mlp.NET_FG_COLOR, 0, W(mlp.NET_FG_COLOR, 0), fr["code_color"])
    ft["pf.color3"] = mlp.pf_bias_of(mlp.NET_FG_COLOR, 3, W(mlp.NET_FG_COLOR, 3), fr["appr_code"])
    ft["pf.vis0"] = mlp.pf_bias_of(mlp.NET_VIS, 0, W(mlp.NET_VIS, 0), fr["code_vis"])
    if fr.get("dense") is None and motion == "skinning":
        t_art, rest = fr["t_articulation"], fr["rest_articulation"]
        skin = mlp.skin_net_for(t_art[0].shape[1])
        ft["gauss"] = get_gauss(P)
        ft["se3_bw.r"], ft["se3_bw.d"] = Q.dual_quaternion_mul(rest, Q.dual_quaternion_inverse(t_art))
        rest_inv = Q.dual_quaternion_inverse(rest)
        ft["se3_next.r"], ft["se3_next.d"] = Q.dual_quaternion_mul(nxt["t_articulation"], rest_inv)
        ft["se3_own.r"], ft["se3_own.d"] = Q.dual_quaternion_mul(t_art, rest_inv)
        ft["pf.skin_bw"] = mlp.pf_bias_of(skin, 0, W(skin, 0), skin_cond(fr["t_embed"], fr["code_skin"], M))
        ft["pf.skin_fw"] = mlp.pf_bias_of(skin, 0, W(skin, 0), skin_cond(fr["t_embed_mean"], fr["code_skin"], M))
        if SKIN_AFFINE:  # the delta-skin MLP's first layer as a per-frame table of the point (warping.skin_affine_table)
            B = t_art[0].shape[1]
            ft["skinA.bw"] = skin_affine_table(P, t_art, ft["gauss"], ft["pf.skin_bw"], B)
            ft["skinA.fw"] = skin_affine_table(P, rest, ft["gauss"], ft["pf.skin_fw"], B)
    return ft


class FramePrologue:
    """The per-frame prologue / epilogue of one training step whose rays are rendered chunk by chunk.

        fr_step = prologue.refresh()        # once per step: frame_terms(P, fr) with its autograd graph; values -> static leaves
        for chunk: render_train(P, fr_step, ...); loss.backward()    # chunks read the leaves, their gradients add up in leaf.grad
        prologue.backward()                 # once per step: the summed leaf gradients flow back to the weights / per-frame inputs

    The leaves keep their addresses across steps, so the chunk can be a captured hipGraph that is replayed."""

    def __init__(self, P, fr):
        self.P, self.fr = P, fr
        self.outs, self.leaves = None, None

    def refresh(self):
        self.outs = frame_terms(self.P, self.fr)
        if self.leaves is None:
            self.leaves = {}
            for k, v in self.outs.items():
                leaf = v.detach().clone().contiguous()
                if v.requires_grad:
                    leaf.requires_grad_(True)
                    leaf.grad = torch.zeros_like(leaf)
                self.leaves[k] = leaf
        else:
            with torch.no_grad():
                for k, v in self.outs.items():
                    self.leaves[k].copy_(v)
        return dict(self.fr, frame_terms=self.leaves)

    def zero_grad(self):
        for leaf in (self.leaves or {}).values():
            if leaf.grad is not None:
                leaf.grad.zero_()

    def backward(self):
        ks = [k for k, v in self.outs.items() if v.requires_grad]
        if ks:
            torch.autograd.backward([self.outs[k] for k in ks], [self.leaves[k].grad for k in ks])
            for k in ks:
                self.leaves[k].grad.zero_()
        self.outs = None


class BgPrologue(FramePrologue):
    """The same once-per-step pattern for the background field of the `comp` configurations: its only ray-independent terms are the
    instance-code lookups (embedding.py:246-264) that feed the per-frame biases."""
    KEYS = {"code_base": "basefield.inst_embedding.mapping.weight", "code_color": "colorfield.inst_embedding.mapping.weight",
            "code_vis": "vis_mlp.basefield.inst_embedding.mapping.weight"}

    def _idx(self, w):
        return self.fr["inst_id"] if w.shape[0] > 1 else torch.zeros_like(self.fr["inst_id"])

    def refresh(self):
        # (no autograd graph: the adjoint of a row lookup is written out in backward() below)
        with torch.no_grad():
            self.outs = {k: self.P[n][self._idx(self.P[n])] for k, n in self.KEYS.items()}
        if self.leaves is None:
            self.leaves = {}
            for k, v in self.outs.items():
                leaf = v.detach().clone().contiguous()
                if self.P[self.KEYS[k]].requires_grad:
                    leaf.requires_grad_(True)
                    leaf.grad = torch.zeros_like(leaf)
                self.leaves[k] = leaf
        else:
            with torch.no_grad():
                for k, v in self.outs.items():
                    self.leaves[k].copy_(v)
        return dict(self.fr, **self.leaves)

    def backward(self):
        """d table[row] += sum of the leaf gradients of the frames that looked the row up -- as an explicit index_add_ (atomic adds, no host
        round trip), not autograd's IndexBackward: that one goes through index_put_(accumulate=True) = a sort + segmented reduction whose capture
        into the whole-step hipGraph took the HIP runtime down (round 5: "capturing the background prologue crashed the runtime"; the comp
        configuration was left on per-chunk graphs for it).  Same values; the instance tables' .grad are views of FlatAdamW's flat gradient."""
        with torch.no_grad():
            for k, n in self.KEYS.items():
                w, g = self.P[n], self.leaves[k].grad
                if g is None or not w.requires_grad:
                    continue
                if w.grad is None:
                    w.grad = torch.zeros_like(w)
                w.grad.index_add_(0, self._idx(w), g)
                g.zero_()
        self.outs = None


def _warp_fn(P, fr, prec):
    """The fg field's warp as a closure over its per-frame inputs: SkinningWarp, or ComposedWarp when fr carries the dense
    post-warp's inputs.  partner=True: the warp into the pair partner's frame (compute_flow, nerf.py:966-973) -- the
    post-warp then sees the partner's time embedding."""
    dense = fr.get("dense")
    motion = fr.get("motion", "skinning")

    def warp(x, t_art, rest_art, t_embed, backward, partner=False):
        if motion == "rigid":  # IdentityWarp (warping.py:59-91)
            return x, {}
        if motion == "dense":  # a bare DenseWarp(D=6) (warping.py:94-170): parameters under "warp.", its own time embedding
            te = flip_pair(fr["t_embed_dense"]) if partner else fr["t_embed_dense"]
            return dense_warp(P, x, te, fr["code_dense_bw" if backward else "code_dense_fw"], backward, prec, prefix="warp.", net=mlp.NET_DENSE6), {}
        if dense is None:
            return skinning_warp(P, x, t_art, rest_art, t_embed, fr["code_skin"], backward, prec)
        d = dict(dense, t_embed=flip_pair(dense["t_embed"])) if partner else dense
        return composed_warp(P, x, t_art, rest_art, t_embed, fr["code_skin"], backward, prec, dense=d)

    return warp


def query_field_train(P, fr, hxy, rng, flow_thresh=None, n_depth=64, alpha=None, prec=mlp.PREC_F32):
    """Training-mode Deformable.query_field (same contract as oracle.lab4d_oracle.query_field_train) for every fg_motion create_warp builds
    on kernels (warping.py:35-48): SkinningWarp ("bob", "skel-*": the default here); with fr["dense"] = {"t_embed", "code_fw", "code_bw"} the
    ComposedWarp of "comp_skel-*_dense" (skinning composed with the dense post-warp, warping.py:445-483); fr["motion"] = "rigid": IdentityWarp
    (the reference's default fg_motion); fr["motion"] = "dense": a bare DenseWarp (fr["t_embed_dense"], fr["code_dense_fw" / "_bw"])."""
    warp = _warp_fn(P, fr, prec)
    ft = fr.get("frame_terms")  # the step's prologue (FramePrologue.refresh), or evaluated here: same arithmetic either way
    if ft is None:
        ft = frame_terms(P, fr)
    motion = fr.get("motion", "skinning")  # "rigid" / "dense": fg_motion rigid (IdentityWarp, the reference's default) / dense (bare DenseWarp)
    skinning = fr.get("dense") is None and motion == "skinning"
    cam2field = (ft["cam2field.q"], ft["cam2field.t"])
    xyz_cam, dir_cam, deltas, depth, xyz_t, _ = RU.ray_samples(hxy, fr["Kinv"], fr["near_far"], cam2field, n_depth=n_depth)
    t_art, rest_art = fr.get("t_articulation"), fr.get("rest_articulation")  # absent for fg_motion "rigid" / "dense"
    if skinning:
        xyz, bw_aux = skinning_warp(P, xyz_t, t_art, rest_art, fr["t_embed"], fr["code_skin"], True, prec,
                                    pre={"se3": (ft["se3_bw.r"], ft["se3_bw.d"]), "gauss": ft["gauss"], "pf": ft["pf.skin_bw"], "tab": ft.get("skinA.bw")})
    else:
        xyz, bw_aux = warp(xyz_t, t_art, rest_art, fr.get("t_embed"), True)
    matching = fr.get("feature") is not None  # FeatureNeRF.query_field (feature.py:104-107): the matching terms need the pixel features of the batch
    if matching:  # the drawn candidates' rows of the canonical points (feature.py:176-178); every consumer below reads the tapped tensor
        xyz_flat, xyz_rows = RowTap.apply(xyz.reshape(-1, 3), rng["match_perm"])
        xyz = xyz_flat.view(xyz.shape)
    fd = {}
    vis = vis_field(P, xyz, fr, prec, ft)
    base_tap = {} if EIK_REUSE else None  # the basefield pass's ReLU pattern, handed to the eikonal term below
    rgb, density = nerf_forward(P, xyz, fr, prec, alpha=alpha, ft=ft, tap=base_tap)
    fd["rgb"], fd["density"], fd["density_fg"], fd["vis"] = rgb, density, density, vis
    # flow: canonical points into the pair partner's camera (nerf.py:948-997)
    nxt = flip_pair({k: fr.get(k) for k in ["Kinv", "field2cam", "t_articulation", "rest_articulation"]})
    # pair partners are frames of one video, so the rest articulation (a per-instance quantity from get_vals_and_mean) equals its
    # flip_pair; a caller that supplies its own states it (patch._frames checks it), anything else takes the general path
    shared = skinning and fr.get("rest_shared_in_pair", True)
    fw_pre = {"gauss": ft["gauss"], "pf": ft["pf.skin_fw"], "tab": ft.get("skinA.fw")} if skinning else None
    if shared:
        # both forward warps of the canonical samples (into the partner's frame here, into the own frame for the cycle term
        # below) see the same skinning field: the rest articulation is a per-instance quantity and pair partners are frames
        # of one video (rest_articulation == its flip_pair), so it is evaluated once (warping.skinning_warp_forward_multi)
        (xyz_next, _), (xyz_cyc, cyc_aux) = skinning_warp_forward_multi(P, xyz, [nxt["t_articulation"], t_art], rest_art,
                                                                        fr["t_embed_mean"], fr["code_skin"], prec,
                                                                        pre=dict(fw_pre, se3s=[(ft["se3_next.r"], ft["se3_next.d"]), (ft["se3_own.r"], ft["se3_own.d"])]))
    else:
        xyz_next, _ = warp(xyz, nxt["t_articulation"], nxt["rest_articulation"], fr.get("t_embed_mean"), False, partner=True)
    # cycle consistency (deformable.py:173-198)
    if not shared:
        xyz_cyc, cyc_aux = warp(xyz, t_art, rest_art, fr.get("t_embed_mean"), False)
    # projection into the partner's camera, flow, validity and the cycle distance: one kernel each way (csrc/flow.hip)
    fd["flow"], fd["cyc_dist"] = FlowCyc.apply(xyz_next, ft["field2cam_next.q"], ft["field2cam_next.t"], ft["Kmat_next"], hxy, xyz_cyc, xyz_t, flow_thresh)
    for k in ["skin_entropy", "delta_skin"]:
        # NeRF.cycle_loss's zeros (nerf.py:905-927) unless the warp reports the term (nerf.py:658-664)
        fd[k] = (cyc_aux[k] + bw_aux[k]) / 2 if k in cyc_aux else torch.zeros_like(fd["cyc_dist"])
    fd["eikonal"] = eikonal_subsample(P, xyz, fr["code_base"], rng.get("eik_inds"), alpha, prec, pf_tables=(ft["pf.base0"], ft["pf.base4"]), tap=base_tap)
    fd["xyz"] = xyz
    fd["xyz_cam"] = xyz_cam
    fd["depth"] = depth / ft["scale"]
    fd["feature"] = compute_feat(P, xyz, prec)
    aux = {}
    has_gauss = motion not in ("rigid", "dense")
    if not matching:
        if has_gauss:
            fd["gauss_density"] = gauss_density(P, xyz, rest_art, ft)
        return fd, deltas, aux
    feat_flat, feat_rows = RowTap.apply(fd["feature"].reshape(-1, fd["feature"].shape[-1]), rng["match_perm"])
    fd["feature"] = feat_flat.view(fd["feature"].shape)
    xyz_matches = global_match(P, fr["feature"], feat_rows, xyz_rows)
    if skinning:
        xm_next, _ = skinning_warp(P, xyz_matches[:, :, None], t_art, rest_art, fr["t_embed_mean"], fr["code_skin"], False,
                                   prec, pre=dict(fw_pre, se3=(ft["se3_own.r"], ft["se3_own.d"])))
    else:
        xm_next, _ = warp(xyz_matches[:, :, None], t_art, rest_art, fr.get("t_embed_mean"), False)
    xyz_reproj = rigid_apply(fr["field2cam"][0], fr["field2cam"][1], xm_next)[:, :, 0]
    aux["xyz_matches"] = xyz_matches
    aux["xyz_reproj"] = xyz_reproj
    aux["xy_reproj"] = pinhole_projection(ft["Kmat"], xyz_reproj)[..., :2]
    if has_gauss:
        fd["gauss_density"] = gauss_density(P, xyz, rest_art, ft)
    return fd, deltas, aux


def render_train(P, fr, hxy, rng, flow_thresh=None, n_depth=64, alpha=None, prec=mlp.PREC_F32):
    """dvr_model.render_samples for field_type == "fg" (engine/model.py:328-361)."""
    fd, deltas, aux = query_field_train(P, fr, hxy, rng, flow_thresh, n_depth, alpha, prec)
    rendered = RU.render_pixel(fd, deltas)
    aux_fg = dict(aux)
    aux_fg.update(rendered)  # one field: the per-category render equals the composite (model.py:346-352)
    rendered = dict(rendered)
    rendered["xyz_matches"] = aux["xyz_matches"]
    rendered["xyz_reproj"] = aux["xyz_reproj"]
    return {"rendered": rendered, "aux_dict": {"fg": aux_fg}}


# ---------------------------------------------------------------------------------------------------
# losses (engine/model.py:401-611), per-ray element-wise glue
# ---------------------------------------------------------------------------------------------------
def mask_balance_wt(mask, vis2d, is_detected):
    """dvr_model.get_mask_balance_wt (model.py:401-424) without host synchronisation (branch -> torch.where)."""
    mask = mask.float()
    vis2d = vis2d.float() * is_detected.float()[:, None, None]
    vis = (vis2d > 0).float()
    npos, nneg = (mask * vis).sum(), ((1 - mask) * vis).sum()
    pos = vis2d.sum() / npos.clamp_min(1e-12)
    neg = vis2d.sum() / nneg.clamp_min(1e-12)
    wt = 0.5 * pos * mask + 0.5 * neg * (1 - mask)
    both = (mask.sum() > 0) & ((1 - mask).sum() > 0)
    return torch.where(both, wt, torch.ones_like(wt))


LOSS_TERMS = ["mask", "feature", "feat_reproj", "rgb", "depth", "flow", "vis", "reg_gauss_mask", "reg_eikonal", "reg_deform_cyc", "reg_delta_skin",
              "reg_skin_entropy"]


class _LossInputs(ctypes.Structure):
    _fields_ = [(n, vp) for n in ("mask", "feature", "xy_reproj", "rgb", "depth", "flow", "vis", "gauss_mask", "eikonal", "cyc_dist", "delta_skin",
                                  "skin_entropy", "t_mask", "t_feature", "t_hxy", "t_rgb", "t_depth", "t_flow", "t_flow_uct", "t_vis2d", "t_detected",
                                  "balance_wt")] + [("hxy_ld", ci), ("dense_uses_mask", ci), ("mask_all", vp), ("vis_bg", vp), ("vis_bg_wt", cf)]


class _LossGrads(ctypes.Structure):
    _fields_ = [(n, vp) for n in ("mask", "feature", "xy_reproj", "rgb", "depth", "flow", "vis", "gauss_mask", "eikonal", "cyc_dist", "delta_skin",
                                  "skin_entropy", "mask_all", "vis_bg")]


_lib.register("lab4d_ray_losses_forward", [ctypes.POINTER(_LossInputs), ci, ci, ctypes.POINTER(cf * 12), vp, vp, vp])
_lib.register("lab4d_ray_losses_backward", [ctypes.POINTER(_LossInputs), ci, ci, ctypes.POINTER(cf * 12), vp, vp, ctypes.POINTER(_LossGrads), vp])
_RENDERED = ("mask", "feature", "xy_reproj", "rgb", "depth", "flow", "vis", "gauss_mask", "eikonal", "cyc_dist", "delta_skin", "skin_entropy")
_TARGETS = ("t_mask", "t_feature", "t_hxy", "t_rgb", "t_depth", "t_flow", "t_flow_uct", "t_vis2d", "t_detected", "balance_wt")


class RayLosses(Function):
    """All per-ray loss terms in one pass each way (include/lab4d_loss.h).  apply(N, weights (12 floats), *rendered (12), *targets
    (10), mask_all, vis_bg) -> (13,): the weighted terms in LOSS_TERMS order, then their total.  mask_all / vis_bg (None for field_type
    "fg") select the comp variant: opaque-composite term, background visibility at 1 %, dense terms masked by vis2d only."""

    @staticmethod
    def _inputs(rendered, targets, extras, hxy_ld):
        a = _LossInputs()
        for n, t in zip(_RENDERED + _TARGETS, rendered + targets):
            setattr(a, n, None if t is None else _lib.dp(t))
        comp = extras[0] is not None or extras[1] is not None
        a.hxy_ld, a.dense_uses_mask = hxy_ld, 0 if comp else 1
        a.mask_all = None if extras[0] is None else _lib.dp(extras[0])
        a.vis_bg = None if extras[1] is None else _lib.dp(extras[1])
        a.vis_bg_wt = 0.01  # model.py:489
        return a

    @staticmethod
    def forward(ctx, N, weights, *tensors):
        rendered = [None if t is None else t.contiguous().float() for t in tensors[:12]]
        targets = [None if t is None else t.contiguous().float() for t in tensors[12:22]]
        extras = [None if t is None else t.contiguous().float() for t in tensors[22:24]]
        _lib.require_device(*[t for t in rendered + targets + extras if t is not None])
        ref = next(t for t in rendered if t is not None)
        R = targets[_TARGETS.index("t_vis2d")].numel()
        hxy_ld = targets[2].shape[-1] if targets[2] is not None else 0
        a = RayLosses._inputs(rendered, targets, extras, hxy_ld)
        w = (cf * 12)(*[float(x) for x in weights])
        acc = torch.empty(24, device=ref.device)
        loss = torch.empty(13, device=ref.device)
        _lib.check(_lib.lib().lab4d_ray_losses_forward(ctypes.byref(a), R, int(N), ctypes.byref(w), _lib.ptr(acc), _lib.ptr(loss), _lib.stream()),
                   "ray_losses_forward")
        ctx.keep = (rendered, targets, extras, acc, [float(x) for x in weights], R, int(N), hxy_ld)
        ctx.shapes = [None if t is None else t.shape for t in tensors[:12]] + [None if t is None else t.shape for t in tensors[22:24]]
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        rendered, targets, extras, acc, weights, R, N, hxy_ld = ctx.keep
        a = RayLosses._inputs(rendered, targets, extras, hxy_ld)
        g_loss = (g[:12] + g[12]).contiguous()
        gr = _LossGrads()
        outs = []
        for i, (n, t) in enumerate(zip(_RENDERED, rendered)):
            o = torch.empty_like(t) if (t is not None and ctx.needs_input_grad[2 + i]) else None
            outs.append(o)
            setattr(gr, n, None if o is None else _lib.dp(o))
        for i, (n, t) in enumerate(zip(("mask_all", "vis_bg"), extras)):
            o = torch.empty_like(t) if (t is not None and ctx.needs_input_grad[24 + i]) else None
            outs.append(o)
            setattr(gr, n, None if o is None else _lib.dp(o))
        w = (cf * 12)(*weights)
        _lib.check(_lib.lib().lab4d_ray_losses_backward(ctypes.byref(a), R, N, ctypes.byref(w), _lib.ptr(acc), _lib.ptr(g_loss), ctypes.byref(gr),
                                                        _lib.stream()), "ray_losses_backward")
        gs = [None if o is None else o.view(sh) for o, sh in zip(outs, ctx.shapes)]
        return (None, None) + tuple(gs[:12]) + (None,) * 10 + tuple(gs[12:])


class LossDict(dict):
    """{term: scalar}; `.total` is their sum, formed by the kernel (summing the dict's values costs one launch per term)."""
    total = None
    vec = None  # the kernel's output: the 12 weighted terms in LOSS_TERMS order, then their total


def losses_fg(results, batch, train_res, weights):
    """compute_recon_loss + mask_losses + rendered regularisers + apply_loss_weights for field_type "fg" (engine/model.py:
    401-611): every term's masked sum / positive count in one kernel pass over the rays, the gradients of all rendered
    inputs in one more (csrc/losses.hip).  `batch["mask_balance_wt"]` (get_mask_balance_wt, model.py:401-424, a function of
    the targets only) is used when the caller precomputed it."""
    r, a = results["rendered"], results["aux_dict"]["fg"]
    bal = batch.get("mask_balance_wt")
    if bal is None:
        bal = mask_balance_wt(batch["mask"], batch["vis2d"], batch["is_detected"])
    wt = []
    for k in LOSS_TERMS:
        w = 1.0 if weights is None or k + "_wt" not in weights else float(weights[k + "_wt"])
        wt.append(w / train_res if k in ("flow", "feat_reproj") else w)
    rendered = [r["mask"], a.get("feature"), a.get("xy_reproj"), r["rgb"], r["depth"], r["flow"], a["vis"], a.get("gauss_mask"), r["eikonal"], a["cyc_dist"],
                a["delta_skin"], a["skin_entropy"]]
    targets = [batch["mask"], batch.get("feature"), batch["hxy"], batch["rgb"], batch["depth"], batch["flow"], batch["flow_uct"], batch["vis2d"],
               batch["is_detected"], bal]
    vec = RayLosses.apply(r["mask"].shape[1], wt, *rendered, *targets, None, None)
    # a term whose rendered input does not exist is not in the reference's loss_dict either (reg_gauss_mask without a SkinningWarp,
    # model.py:497-501; the matching terms without pixel features); the kernel reports it as 0 in `vec` / `total`
    out = LossDict((k, vec[i]) for i, k in enumerate(LOSS_TERMS) if rendered[i] is not None)
    out.total, out.vec = vec[12], vec
    return out


def losses_fg_reference_ops(results, batch, train_res, weights):
    """The same terms written with element-wise tensor operations, one term at a time like the reference -- kept as the
    readable statement of what RayLosses computes (tests compare the two); not used by the renderer."""
    r, a = results["rendered"], results["aux_dict"]["fg"]
    L = {}
    L["mask"] = (r["mask"] - batch["mask"].float()).pow(2) * mask_balance_wt(batch["mask"], batch["vis2d"], batch["is_detected"])
    L["feature"] = (a["feature"] - batch["feature"]).norm(2, -1, keepdim=True)
    L["feat_reproj"] = (a["xy_reproj"] - batch["hxy"][..., :2]).norm(2, -1, keepdim=True)
    L["rgb"] = (r["rgb"] - batch["rgb"]).pow(2)
    L["depth"] = (r["depth"] - batch["depth"]).norm(2, -1, keepdim=True)
    L["flow"] = (r["flow"] - batch["flow"]).norm(2, -1, keepdim=True) * (batch["flow_uct"] > 0).float()
    L["vis"] = a["vis"]
    L["reg_gauss_mask"] = (a["gauss_mask"] - r["mask"].detach()).pow(2)
    vis2d, mfg = batch["vis2d"].float(), batch["mask"].float()
    det = batch["is_detected"].float()[:, None, None]
    for k in list(L.keys()):
        if k == "reg_gauss_mask":
            continue
        if k == "mask":
            L[k] = L[k] * vis2d
        elif k in ("feature", "feat_reproj"):
            L[k] = L[k] * mfg
        else:
            L[k] = L[k] * (mfg * vis2d)
        if k in ("mask", "feature", "feat_reproj"):
            L[k] = L[k] * det
    L["reg_eikonal"] = r["eikonal"]
    L["reg_deform_cyc"] = a["cyc_dist"]
    L["reg_delta_skin"] = a["delta_skin"]
    L["reg_skin_entropy"] = a["skin_entropy"]
    out = {}
    for k, v in L.items():
        pos = (v > 0).to(v.dtype)
        v = (v * pos).sum() / pos.sum()  # == v[v > 0].mean() (model.py:602) without the host sync of boolean indexing
        if k in ("flow", "feat_reproj"):
            v = v / train_res
        if weights is not None and k + "_wt" in weights:
            v = v * weights[k + "_wt"]
        out[k] = v
    return out


# lab4d/config.py:10-46 defaults
DEFAULT_LOSS_WT = {
    "mask_wt": 0.1, "rgb_wt": 0.1, "depth_wt": 1e-4, "flow_wt": 0.5, "vis_wt": 1e-2, "feature_wt": 1e-2, "feat_reproj_wt": 5e-2,
    "reg_visibility_wt": 1e-4, "reg_eikonal_wt": 1e-3, "reg_deform_cyc_wt": 0.01, "reg_delta_skin_wt": 5e-3,
    "reg_skin_entropy_wt": 5e-4, "reg_gauss_skin_wt": 1e-3, "reg_cam_prior_wt": 0.1, "reg_skel_prior_wt": 0.1,
    "reg_gauss_mask_wt": 0.01, "reg_soft_deform_wt": 100.0,
}


# ---------------------------------------------------------------------------------------------------
# evaluation graph: importance sampling, valid-mask, normals   (nerf.py:455-528, 605-636, 686-738, 769-819)
# ---------------------------------------------------------------------------------------------------
def extend_aabb(aabb, factor=0.1):
    """geom_utils.py:409-422."""
    ext = (aabb[1] - aabb[0]) * factor
    return torch.stack([aabb[0] - ext, aabb[1] + ext], 0)


def check_inside_aabb(xyz, aabb):
    """geom_utils.py:506-517 (strict inequalities)."""
    return ((xyz > aabb[:1]) & (xyz < aabb[1:])).all(-1)


def get_valid_idx(P, xyz, xyz_t, t_articulation):
    """NeRF.get_valid_idx (nerf.py:495-528): bool (M,N,D)."""
    valid = check_inside_aabb(xyz, extend_aabb(P["aabb"]))
    _, tb = Q.dual_quaternion_to_quaternion_translation(t_articulation)
    tb = tb[0]
    t_aabb = extend_aabb(torch.stack([tb.min(0)[0], tb.max(0)[0]], 0), factor=1.0)
    return valid & check_inside_aabb(xyz_t, t_aabb)


@torch.no_grad()
def importance_sampling(P, fr, hxy, n_depth=64, alpha=None, prec=mlp.PREC_F32):
    """NeRF.importance_sampling (nerf.py:686-738): n/2 uniform samples -> density -> weights -> inverse-CDF samples
    (sample_pdf, det=True) -> merge-sort -> 64 depths.  Returns (xyz_cam, dir_cam, deltas, depth, xyz_t, .), (inds, coarse weights)."""
    nc = n_depth // 2
    cam2field = Q.quaternion_translation_inverse(fr["field2cam"][0], fr["field2cam"][1])
    xyz_cam, _, deltas, depth, xyz_t, _ = RU.ray_samples(hxy, fr["Kinv"], fr["near_far"], cam2field, n_depth=nc)
    xyz, _ = _warp_fn(P, fr, prec)(xyz_t, fr.get("t_articulation"), fr.get("rest_articulation"), fr.get("t_embed"), True)
    density = nerf_forward(P, xyz, fr, prec, with_color=False, alpha=alpha)
    weights, _ = RU.compute_weights(density, deltas)
    depth_mid = (0.5 * (depth[:, :, :-1] + depth[:, :, 1:])).reshape(-1, nc - 1)
    new, inds = RU.sample_pdf(depth_mid, weights.reshape(-1, nc)[:, 1:-1].contiguous(), nc, det=True, return_inds=True)
    depth_all = RU.sort_depth(depth.reshape(-1, nc), new).view(depth.shape[0], depth.shape[1], n_depth, 1)
    return RU.ray_samples(hxy, fr["Kinv"], fr["near_far"], cam2field, depth=depth_all), (inds, weights)


def nerf_forward_compacted(P, x_k, fr, frame_k, count, prec, alpha=None):
    """NeRF.forward on the stream-compacted valid samples (NeRF.query_nerf, nerf.py:782-808): x_k (S,3) with the valid samples
    in its first *count rows, frame_k (S) int32 their frames.  Returns (rgb (S,3), density (S,1)); rows >= *count are undefined."""
    dev = x_k.device
    sdf, feat = mlp.run_chain_compacted(mlp.NET_FG_BASE, prec, P, x_k, frame_k, count, conds={0: fr["code_base"], 4: fr["code_base"]},
                                        export_layer=8, freq_w=posenc_window(alpha, 10, dev))
    density = volsdf_density(sdf, P["logibeta"])
    rgb = mlp.run_chain_compacted(mlp.NET_FG_COLOR, prec, P, x_k, frame_k, count, conds={0: fr["code_color"], 3: fr["appr_code"]}, ext=feat,
                                  freq_w=posenc_window(alpha, 12, dev))
    return torch.sigmoid(rgb), density


def query_field_eval(P, fr, hxy, n_depth=64, alpha=None, prec=mlp.PREC_F32):
    """Eval-mode Deformable.query_field (train-only fields return {}, decorator.py:4-17): importance sampling, backward warp,
    visibility, get_valid_idx + query_nerf (the field's colour / density only on the VALID samples: mask, stream compaction and
    scatter on the device, no host synchronisation -- nerf.py:495-528, 769-819), normals / eikonal on every sample
    (compute_normal, nerf.py:455-493)."""
    (xyz_cam, dir_cam, deltas, depth, _, _), (inds, weights_coarse) = importance_sampling(P, fr, hxy, n_depth, alpha, prec)
    # normals need d sdf / d xyz_cam through the rigid transform and the warp (nerf.py:455-493): one first-order
    # backward pass through the same kernels, so the sdf pass below is run under autograd with xyz_cam as the leaf.
    # Only d sdf / d xyz_cam is wanted: parameters and per-frame inputs are detached, otherwise the backward below would
    # also run every weight-gradient kernel (needs_input_grad follows requires_grad, not what autograd.grad asked for)
    det = lambda v: tuple(t.detach() for t in v) if isinstance(v, tuple) else (v.detach() if torch.is_tensor(v) else v)
    P = {k: det(v) for k, v in P.items()}
    fr = {k: (det(v) if not isinstance(v, dict) else {a: det(b) for a, b in v.items()}) for k, v in fr.items()}
    with torch.enable_grad():
        xc = xyz_cam.detach().requires_grad_(True)
        qi, ti = Q.quaternion_translation_inverse(fr["field2cam"][0], fr["field2cam"][1])
        xyz_t = rigid_apply(qi, ti, xc)
        xyz, _ = _warp_fn(P, fr, prec)(xyz_t, fr.get("t_articulation"), fr.get("rest_articulation"), fr.get("t_embed"), True)
        sdf = nerf_forward(P, xyz, fr, prec, with_color=False, get_density=False, alpha=alpha)
        (g,) = torch.autograd.grad(sdf, xc, torch.ones_like(sdf))
    with torch.no_grad():
        xyz, xyz_t = xyz.detach(), xyz_t.detach()
        shape = xyz.shape[:-1]
        S, spf = xyz.numel() // 3, _spf(xyz)
        vis = vis_field(P, xyz, fr, prec)
        # get_valid_idx (nerf.py:495-528): inside the field's aabb (+10 %) and -- when the samples carry articulations, i.e. for SkinningWarp
        # fields (deformable.py:254-289) -- in time-t space, inside the box of frame 0's bones (x2)
        has_bones = fr.get("motion", "skinning") not in ("rigid", "dense")
        t_aabb = None
        if has_bones:
            _, tb = Q.dual_quaternion_to_quaternion_translation(fr["t_articulation"])
            t_aabb = extend_aabb(torch.stack([tb[0].min(0)[0], tb[0].max(0)[0]], 0), factor=1.0)
        mask = RU.valid_mask(xyz, xyz_t, extend_aabb(P["aabb"]), t_aabb)
        # query_nerf (nerf.py:782-819): the field on the valid samples only, scattered into zeros
        idx, count = RU.compact(mask)
        x_k = RU.gather_rows(xyz.reshape(-1, 3), idx, count)
        rgb_k, dens_k = nerf_forward_compacted(P, x_k, fr, RU.frame_of(idx, count, spf), count, prec, alpha)
        rgb = RU.scatter_rows(rgb_k, idx, count, S).view(shape + (3,))
        density = RU.scatter_rows(dens_k, idx, count, S).view(shape + (1,))
        fd = {"rgb": rgb, "density": density, "density_fg": density, "vis": vis}
        fd["eikonal"] = (g.norm(2, dim=-1, keepdim=True) - 1) ** 2
        fd["normal"] = _flip_yz(F.normalize(g, dim=-1))
        fd["xyz"] = xyz
        fd["xyz_cam"] = xyz_cam
        fd["depth"] = depth / P["logscale"].exp()
        if has_bones:
            fd["gauss_density"] = gauss_density(P, xyz, fr["rest_articulation"])
    # debug outputs (the parity tests read them): sample_pdf's indices, the coarse pass's compositing weights its pdf was formed from, the valid mask
    return fd, deltas, {"valid": mask.view(shape).bool(), "inds": inds, "weights_coarse": weights_coarse, "valid_count": count}


@torch.no_grad()
def render_eval(P, fr, hxy, n_depth=64, alpha=None, prec=mlp.PREC_F32):
    """dvr_model.render (eval) for field_type == "fg"."""
    fd, deltas, dbg = query_field_eval(P, fr, hxy, n_depth, alpha, prec)
    out = RU.render_pixel(fd, deltas)
    return {"rendered": out, "aux_dict": {"fg": dict(out)}, "debug": dbg}


# ---------------------------------------------------------------------------------------------------
# background field + fg/bg composition in eval mode  (nerf.py:580-684 for category "bg"; engine/model.py:328-361, "comp")
# ---------------------------------------------------------------------------------------------------
@torch.no_grad()
def importance_sampling_bg(P, fr, hxy, n_depth=64, alpha=None, prec=mlp.PREC_F32, prefix=""):
    """NeRF.importance_sampling (nerf.py:686-738) with the rigid backward warp of the background field.  Returns the samples and (inds, coarse weights)."""
    nc = n_depth // 2
    codes = {"basefield": fr["code_base"], "colorfield": fr["code_color"]}
    cam2field = Q.quaternion_translation_inverse(fr["field2cam"][0], fr["field2cam"][1])
    _, _, deltas, depth, xyz, _ = RU.ray_samples(hxy, fr["Kinv"], fr["near_far"], cam2field, n_depth=nc)
    density = nerf_forward_bg(P, xyz, None, codes, prec, alpha=alpha, prefix=prefix)
    weights, _ = RU.compute_weights(density, deltas)
    depth_mid = (0.5 * (depth[:, :, :-1] + depth[:, :, 1:])).reshape(-1, nc - 1)
    new, inds = RU.sample_pdf(depth_mid, weights.reshape(-1, nc)[:, 1:-1].contiguous(), nc, det=True, return_inds=True)
    depth_all = RU.sort_depth(depth.reshape(-1, nc), new).view(depth.shape[0], depth.shape[1], n_depth, 1)
    return RU.ray_samples(hxy, fr["Kinv"], fr["near_far"], cam2field, depth=depth_all), (inds, weights)


def query_field_eval_bg(P, fr, hxy, n_depth=64, alpha=None, prec=mlp.PREC_F32, prefix=""):
    """Eval-mode NeRF.query_field of the background field: rigid warp, no valid-index compaction (get_valid_idx returns
    None for category "bg", nerf.py:524-526), zero cycle terms (nerf.py:905-925), normals through the rigid transform."""
    (xyz_cam, dir_cam, deltas, depth, _, dir_f), (inds, weights_coarse) = importance_sampling_bg(P, fr, hxy, n_depth, alpha, prec, prefix)
    det = lambda v: tuple(t.detach() for t in v) if isinstance(v, tuple) else (v.detach() if torch.is_tensor(v) else v)
    P = {k: det(v) for k, v in P.items()}
    fr = {k: det(v) for k, v in fr.items()}
    codes = {"basefield": fr["code_base"], "colorfield": fr["code_color"]}
    with torch.enable_grad():
        xc = xyz_cam.detach().requires_grad_(True)
        qi, ti = Q.quaternion_translation_inverse(fr["field2cam"][0], fr["field2cam"][1])
        xyz = rigid_apply(qi, ti, xc)
        rgb, sdf = nerf_forward_bg(P, xyz, dir_f.detach(), codes, prec, get_density=False, alpha=alpha, prefix=prefix)
        (g,) = torch.autograd.grad(sdf, xc, torch.ones_like(sdf))
    with torch.no_grad():
        xyz, rgb, sdf = xyz.detach(), rgb.detach(), sdf.detach()
        ibeta = P[prefix + "logibeta"].exp()
        density = (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() * ibeta)) * ibeta
        fd = {"rgb": rgb, "density": density, "density_bg": density, "vis": vis_field(P, xyz, fr, prec)}
        for k in ("cyc_dist", "delta_skin", "skin_entropy"):
            fd[k] = torch.zeros_like(density)
        fd["eikonal"] = (g.norm(2, dim=-1, keepdim=True) - 1) ** 2
        fd["normal"] = _flip_yz(F.normalize(g, dim=-1))
        fd["xyz"] = xyz
        fd["xyz_cam"] = xyz_cam
        fd["depth"] = depth / P[prefix + "logscale"].exp()
    return fd, deltas, {"inds": inds, "weights_coarse": weights_coarse}


def query_field_train_bg(P, fr, hxy, rng, flow_thresh=None, n_depth=64, alpha=None, prec=mlp.PREC_F32, prefix=""):
    """Training-mode NeRF.query_field of the background field (nerf.py:580-684): rigid warps, flow into the pair partner's
    camera (nerf.py:948-997), zero cycle terms (nerf.py:905-925), eikonal on the host-drawn ray subset (same contract as
    oracle.lab4d_oracle.query_field_train_bg)."""
    codes = {"basefield": fr["code_base"], "colorfield": fr["code_color"]}
    cam2field = Q.quaternion_translation_inverse(fr["field2cam"][0], fr["field2cam"][1])
    xyz_cam, _, deltas, depth, xyz, dirs = RU.ray_samples(hxy, fr["Kinv"], fr["near_far"], cam2field, n_depth=n_depth)
    fd = {}
    vis = vis_field(P, xyz, fr, prec)
    base_tap = {} if EIK_REUSE else None  # as in query_field_train: the eikonal term takes its primal ReLU pattern from this pass
    rgb, density = nerf_forward_bg(P, xyz, dirs, codes, prec, alpha=alpha, prefix=prefix, tap=base_tap)
    fd["rgb"], fd["density"], fd["density_bg"], fd["vis"] = rgb, density, density, vis
    nxt = flip_pair({k: fr[k] for k in ["Kinv", "field2cam"]})
    fd["flow"], _ = FlowCyc.apply(xyz, nxt["field2cam"][0], nxt["field2cam"][1], Q.kmatinv(nxt["Kinv"]), hxy, None, None, flow_thresh)
    for k in ("cyc_dist", "delta_skin", "skin_entropy"):
        fd[k] = torch.zeros_like(density)
    fd["eikonal"] = eikonal_subsample(P, xyz, fr["code_base"], rng.get("eik_inds_bg"), alpha, prec, net=mlp.NET_BG_BASE, prefix=prefix, tap=base_tap)
    fd["xyz"] = xyz
    fd["xyz_cam"] = xyz_cam
    fd["depth"] = depth / P[prefix + "logscale"].exp()
    return fd, deltas, {}


def render_train_comp(P_fg, fr_fg, P_bg, fr_bg, hxy, rng, flow_thresh=None, n_depth=64, alpha=None, prec=mlp.PREC_F32):
    """dvr_model.render_samples for field_type == "comp" in training mode (engine/model.py:328-361)."""
    from . import multifields
    fd_fg, d_fg, aux = query_field_train(P_fg, fr_fg, hxy, rng, flow_thresh, n_depth, alpha, prec)
    fd_bg, d_bg, _ = query_field_train_bg(P_bg, fr_bg, hxy, rng, flow_thresh, n_depth, alpha, prec)
    fd, deltas = multifields.compose_fields({"fg": fd_fg, "bg": fd_bg}, {"fg": d_fg, "bg": d_bg})
    rendered = dict(RU.render_pixel(fd, deltas))
    aux_fg = dict(aux)
    aux_fg.update(RU.render_pixel(fd_fg, d_fg))
    rendered["xyz_matches"] = aux["xyz_matches"]
    rendered["xyz_reproj"] = aux["xyz_reproj"]
    return {"rendered": rendered, "aux_dict": {"fg": aux_fg, "bg": dict(RU.render_pixel(fd_bg, d_bg))}}


def losses_comp(results, batch, train_res, weights):
    """compute_recon_loss + mask_losses + rendered regularisers + apply_loss_weights for field_type "comp" (model.py:426-611) through
    the same two kernel passes as `losses_fg` (csrc/losses.hip, comp inputs): fg mask = rendered mask_fg, the composite must be
    opaque, visibility supervised per field (bg at 1 %), dense terms masked by vis2d only."""
    r, a, b = results["rendered"], results["aux_dict"]["fg"], results["aux_dict"]["bg"]
    bal = batch.get("mask_balance_wt")
    if bal is None:
        bal = mask_balance_wt(batch["mask"], batch["vis2d"], batch["is_detected"])
    wt = []
    for k in LOSS_TERMS:
        w = 1.0 if weights is None or k + "_wt" not in weights else float(weights[k + "_wt"])
        wt.append(w / train_res if k in ("flow", "feat_reproj") else w)
    rendered = [r["mask_fg"], a["feature"], a["xy_reproj"], r["rgb"], r["depth"], r["flow"], a["vis"], a["gauss_mask"], r["eikonal"], a["cyc_dist"],
                a["delta_skin"], a["skin_entropy"]]
    targets = [batch["mask"], batch["feature"], batch["hxy"], batch["rgb"], batch["depth"], batch["flow"], batch["flow_uct"], batch["vis2d"],
               batch["is_detected"], bal]
    vec = RayLosses.apply(r["mask"].shape[1], wt, *rendered, *targets, r["mask"], b["vis"])
    out = LossDict((k, vec[i]) for i, k in enumerate(LOSS_TERMS))
    out.total, out.vec = vec[12], vec
    return out


def losses_comp_reference_ops(results, batch, train_res, weights):
    """The comp terms written with element-wise tensor operations, one term at a time like the reference -- the readable statement of
    what the kernel computes (tests compare the two); not used by the renderer."""
    r, a, b = results["rendered"], results["aux_dict"]["fg"], results["aux_dict"]["bg"]
    L = {}
    L["mask"] = (r["mask_fg"] - batch["mask"].float()).pow(2) * mask_balance_wt(batch["mask"], batch["vis2d"], batch["is_detected"]) \
        + (r["mask"] - 1).pow(2)
    L["feature"] = (a["feature"] - batch["feature"]).norm(2, -1, keepdim=True)
    L["feat_reproj"] = (a["xy_reproj"] - batch["hxy"][..., :2]).norm(2, -1, keepdim=True)
    L["rgb"] = (r["rgb"] - batch["rgb"]).pow(2)
    L["depth"] = (r["depth"] - batch["depth"]).norm(2, -1, keepdim=True)
    L["flow"] = (r["flow"] - batch["flow"]).norm(2, -1, keepdim=True) * (batch["flow_uct"] > 0).float()
    L["vis"] = a["vis"] + 0.01 * b["vis"]
    L["reg_gauss_mask"] = (a["gauss_mask"] - r["mask_fg"].detach()).pow(2)
    vis2d, mfg = batch["vis2d"].float(), batch["mask"].float()
    det = batch["is_detected"].float()[:, None, None]
    for k in list(L.keys()):
        if k == "reg_gauss_mask":
            continue
        if k == "mask":
            L[k] = L[k] * vis2d
        elif k in ("feature", "feat_reproj"):
            L[k] = L[k] * mfg
        else:
            L[k] = L[k] * vis2d
        if k in ("mask", "feature", "feat_reproj"):
            L[k] = L[k] * det
    L["reg_eikonal"] = r["eikonal"]
    L["reg_deform_cyc"] = a["cyc_dist"]
    L["reg_delta_skin"] = a["delta_skin"]
    L["reg_skin_entropy"] = a["skin_entropy"]
    out = {}
    for k, v in L.items():
        pos = (v > 0).to(v.dtype)
        v = (v * pos).sum() / pos.sum()
        if k in ("flow", "feat_reproj"):
            v = v / train_res
        if weights is not None and k + "_wt" in weights:
            v = v * weights[k + "_wt"]
        out[k] = v
    return out


@torch.no_grad()
def render_eval_comp(P_fg, fr_fg, P_bg, fr_bg, hxy, n_depth=64, alpha=None, prec=mlp.PREC_F32):
    """dvr_model.render_samples for field_type == "comp" in eval mode (engine/model.py:328-361): both fields are queried on
    the same rays, z-merged by MultiFields.compose_fields, and the composite and each field are rendered."""
    from . import multifields
    fd_fg, d_fg, dbg_fg = query_field_eval(P_fg, fr_fg, hxy, n_depth, alpha, prec)
    fd_bg, d_bg, dbg_bg = query_field_eval_bg(P_bg, fr_bg, hxy, n_depth, alpha, prec)
    fd, deltas = multifields.compose_fields({"fg": fd_fg, "bg": fd_bg}, {"fg": d_fg, "bg": d_bg})
    return {"rendered": RU.render_pixel(fd, deltas), "aux_dict": {"fg": RU.render_pixel(fd_fg, d_fg), "bg": RU.render_pixel(fd_bg, d_bg)},
            "composed": fd, "deltas": deltas, "debug": {"fg": dbg_fg, "bg": dbg_bg}}
