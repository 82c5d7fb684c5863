"""Host-side mirror of lab4d/nnutils/warping.py SkinningWarp.forward (+ skinning.py SkinningField,
transforms.get_bone_coords, geom_utils.dual_quaternion_skinning, loss_utils.cross_entropy_skin_loss)
on the gfx950 kernels of csrc/skinning.hip and the fused delta-skin MLP (LAB4D_NET_SKIN)."""
import os
import types

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib, mlp
from . import quat_utils as Q

vp, ci = _lib.vp, _lib.ci
_lib.register("lab4d_bone_coords_forward", [vp] * 4 + [ci] * 4 + [vp, vp])
_lib.register("lab4d_bone_coords_backward", [vp] * 5 + [ci] * 4 + [vp] * 4 + [vp])
_lib.register("lab4d_skin_blend_forward", [vp] * 7 + [ci] * 4 + [vp] * 4 + [vp])
_lib.register("lab4d_skin_blend_backward", [vp] * 10 + [ci] * 4 + [vp] * 7 + [vp])
_lib.register("lab4d_skin_blend_backward_acc", [vp] * 10 + [ci] * 4 + [vp] * 7 + [ci, vp])
_lib.register("lab4d_skin_blend_backward_workspace_floats", [ci] * 5)


def _blend_bwd_work(S, spf, M, B, device):
    """Scratch of the blend adjoint, sized by the library (which also decides fused / unfused: one parse of LAB4D_BLEND_FUSE, include/lab4d_skin.h).
    Returns (work, fused)."""
    n = int(_lib.lib().lab4d_skin_blend_backward_workspace_floats(S, spf, M, B, 1))
    return torch.empty(n, device=device), n == M * B * 34
_lib.register("lab4d_gram_per_frame", [vp, ci, vp, ci, ci, ci, ci, vp, vp])
_lib.register("lab4d_bone_params_from_gram", [vp] * 4 + [ci] * 2 + [vp] * 3 + [vp])
_lib.register("lab4d_bone_affine", [vp] * 3 + [ci] * 2 + [vp, vp])
_lib.register("lab4d_bone_coords_backward_gram", [vp] * 4 + [ci] * 4 + [vp, vp, vp])

# The delta-skin chain forms the bone coordinates in its own kernel (SkinChain); 0 restores the two-kernel form (A/B measurements).
FUSE_BONE_COORDS = os.environ.get("LAB4D_FUSE_BONE", "1") != "0"


class BoneCoords(Function):
    """(S,3) points -> (S,3B) gaussian-scaled bone coordinates (transforms.py:9-25, skinning.py:126-140)."""

    @staticmethod
    def forward(ctx, xyz, art_r, art_d, gauss, spf):
        xyz, art_r, art_d, gauss = xyz.contiguous(), art_r.contiguous(), art_d.contiguous(), gauss.contiguous()
        _lib.require_device(xyz, art_r, art_d, gauss)
        S, (M, B) = xyz.shape[0], art_r.shape[:2]
        out = torch.empty(S, 3 * B, device=xyz.device)
        with _lib.timed("k_bone_fwd", (0.0, 4.0 * S * (3 + 3 * B))):  # algorithmic bytes: xyz read, (S,3B) written
            _lib.check(_lib.lib().lab4d_bone_coords_forward(_lib.ptr(xyz), _lib.ptr(art_r), _lib.ptr(art_d), _lib.ptr(gauss), S, spf, M, B,
                                                            _lib.ptr(out), _lib.stream()), "bone_coords_forward")
        ctx.save_for_backward(xyz, art_r, art_d, gauss)
        ctx.spf = spf
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xyz, art_r, art_d, gauss = ctx.saved_tensors
        S, (M, B) = xyz.shape[0], art_r.shape[:2]
        g = g.contiguous()
        gx = torch.empty_like(xyz)
        need_p = any(ctx.needs_input_grad[1:4])
        if need_p and ctx.spf % 256 == 0 and os.environ.get("LAB4D_BONE_FUSE", "1") != "0":
            # one pass over the (S,3B) gradient: the point gradient and the per-frame Gram matrix of the parameter path together
            G = torch.empty(M, 3 * B, 4, device=xyz.device)
            with _lib.timed("k_bone_bwd_x+gram", (2.0 * S * 3 * B * 4, 4.0 * S * (3 + 3 + 3 * B))):
                _lib.check(_lib.lib().lab4d_bone_coords_backward_gram(_lib.ptr(xyz), _lib.ptr(art_r), _lib.ptr(gauss), _lib.ptr(g), S, ctx.spf, M, B,
                                                                      _lib.ptr(gx), _lib.ptr(G), _lib.stream()), "bone_coords_backward_gram")
            need_r, need_d, need_g = ctx.needs_input_grad[1:4]
            gar = torch.empty_like(art_r) if need_r else None
            gad = torch.empty_like(art_d) if need_d else None
            gg = torch.zeros_like(gauss) if need_g else None
            _lib.check(_lib.lib().lab4d_bone_params_from_gram(_lib.ptr(art_r), _lib.ptr(art_d), _lib.ptr(gauss), _lib.ptr(G), M, B, _lib.ptr(gar),
                                                              _lib.ptr(gad), _lib.ptr(gg), _lib.stream()), "bone_params_from_gram")
            return gx, gar, gad, gg, None
        with _lib.timed("k_bone_bwd_x", (0.0, 4.0 * S * (3 + 3 * B))):
            _lib.check(_lib.lib().lab4d_bone_coords_backward(_lib.ptr(xyz), _lib.ptr(art_r), _lib.ptr(art_d), _lib.ptr(gauss), _lib.ptr(g), S, ctx.spf,
                                                             M, B, _lib.ptr(gx), None, None, None, _lib.stream()), "bone_coords_backward")
        # parameter gradients: out[s,b,:] = (R_b x_s + t_b) / gauss_b is affine in x_s, so every one of them is a function
        # of the per-frame Gram matrix G[m,b,k,j] = sum_{s in m} g[s,b,k] [x_s,1]_j: one tall-skinny product on the device
        if not any(ctx.needs_input_grad[1:4]):
            return gx, None, None, None, None
        xh = torch.cat([xyz, torch.ones_like(xyz[:, :1])], -1)
        G = torch.zeros(M, 3 * B, 4, device=xyz.device)
        with _lib.timed("k_gram_pf_rb(bone)", (2.0 * S * 3 * B * 4, 4.0 * S * (3 * B + 4))):
            _lib.check(_lib.lib().lab4d_gram_per_frame(_lib.ptr(g), 3 * B, _lib.ptr(xh), 4, S, ctx.spf, M, _lib.ptr(G), _lib.stream()), "gram_per_frame")
        # (M,B)-sized chain rule: one thread per (frame, bone) (csrc/skinning.hip k_bone_param_from_gram)
        need_r, need_d, need_g = ctx.needs_input_grad[1:4]
        gar = torch.empty_like(art_r) if need_r else None
        gad = torch.empty_like(art_d) if need_d else None
        gg = torch.zeros_like(gauss) if need_g else None
        if need_r or need_d or need_g:
            _lib.check(_lib.lib().lab4d_bone_params_from_gram(_lib.ptr(art_r), _lib.ptr(art_d), _lib.ptr(gauss), _lib.ptr(G), M, B, _lib.ptr(gar),
                                                              _lib.ptr(gad), _lib.ptr(gg), _lib.stream()), "bone_params_from_gram")
        return gx, gar, gad, gg, None


def bone_affine(art_r, art_d, gauss):
    """(M, 3B, 4) fp32: row 3b+k holds the affine map point -> k-th gaussian-scaled coordinate in bone b (lab4d_bone_affine)."""
    art_r, art_d, gauss = art_r.contiguous(), art_d.contiguous(), gauss.contiguous()
    _lib.require_device(art_r, art_d, gauss)
    M, B = art_r.shape[:2]
    aff = torch.empty(M, 3 * B, 4, device=art_r.device)
    _lib.check(_lib.lib().lab4d_bone_affine(_lib.ptr(art_r), _lib.ptr(art_d), _lib.ptr(gauss), M, B, _lib.ptr(aff), _lib.stream()), "bone_affine")
    return aff


class SkinChain(Function):
    """BoneCoords followed by the delta-skin MLP (skinning.py:89-124) with the (S,3B) bone coordinates never written: the chain
    kernel forms them from the points and the per-frame affine table while it stages its input tile (lab4d_mlp_fwd_args.aff).
    The backward pass is the two existing ones back to back: the chain's input gradient (S,3B) feeds k_bone_bwd_x and the
    per-frame Gram reduction of BoneCoords.backward.  Arguments after gauss are those of mlp.MlpChain after x2."""

    @staticmethod
    def forward(ctx, net, prec, spf, xyz, art_r, art_d, gauss, n_pf, *rest):
        xyz = xyz.contiguous()
        inner = types.SimpleNamespace(needs_input_grad=(False, False, False, True, False, False, False, False, False) + tuple(ctx.needs_input_grad[8:]))
        out = mlp.MlpChain.forward(inner, net, prec, spf, xyz, None, None, -1, n_pf, None, *rest, aff=bone_affine(art_r, art_d, gauss))
        ctx.inner, ctx.spf = inner, spf
        ctx.save_for_backward(xyz, art_r.contiguous(), art_d.contiguous(), gauss.contiguous())
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out):
        res = mlp.MlpChain.backward(ctx.inner, d_out)
        bone_ctx = types.SimpleNamespace(saved_tensors=ctx.saved_tensors, spf=ctx.spf, needs_input_grad=(True,) + tuple(ctx.needs_input_grad[4:7]) + (False,))
        gx, gar, gad, gg, _ = BoneCoords.backward(bone_ctx, res[3])
        ctx.inner = None
        return (None, None, None, gx, gar, gad, gg, None) + tuple(res[9:])


class BoneAffine(Function):
    """bone_affine with its adjoint: dL/d aff (M, 3B, 4) IS the per-frame Gram matrix G[m, 3b+k, j] = sum_s g_coord[s, 3b+k] [x_s; 1]_j that
    BoneCoords.backward reduces from the (S, 3B) coordinate gradient, so the (frame, bone)-sized chain rule to the articulation and the
    gaussian scales is the same kernel (lab4d_bone_params_from_gram)."""

    @staticmethod
    def forward(ctx, art_r, art_d, gauss):
        art_r, art_d, gauss = art_r.contiguous(), art_d.contiguous(), gauss.contiguous()
        ctx.save_for_backward(art_r, art_d, gauss)
        return bone_affine(art_r, art_d, gauss)

    @staticmethod
    @once_differentiable
    def backward(ctx, G):
        art_r, art_d, gauss = ctx.saved_tensors
        M, B = art_r.shape[:2]
        G = G.contiguous()
        need_r, need_d, need_g = ctx.needs_input_grad
        gar = torch.empty_like(art_r) if need_r else None
        gad = torch.empty_like(art_d) if need_d else None
        gg = torch.zeros_like(gauss) if need_g else None
        if need_r or need_d or need_g:
            _lib.check(_lib.lib().lab4d_bone_params_from_gram(_lib.ptr(art_r), _lib.ptr(art_d), _lib.ptr(gauss), _lib.ptr(G), M, B, _lib.ptr(gar),
                                                              _lib.ptr(gad), _lib.ptr(gg), _lib.stream()), "bone_params_from_gram")
        return gar, gad, gg


def skin_affine_table(P, art, gauss, pf, n_bones):
    """(M, 64, 4): the delta-skin MLP's first layer as a per-frame affine map of the POINT.  SkinningField.forward feeds linear_1 the
    gaussian-scaled bone coordinates through PosEmbedding(3B, num_freq=0) = identity (skinning.py:69,107), and the coordinates are affine in
    the point per (frame, bone) (transforms.py:9-25): z0 = W1[:, :3B] (aff[m] [x; 1]) + pf[m] + b1 = Wf[m] [x; 1].  Frame-sized torch algebra:
    autograd carries dL/dWf (reduced per frame inside the backward chain kernel) back to linear_1, the articulation, the gaussian scales and
    the conditioning codes.  pf (M, 64) = [time embedding | instance code] @ W1[:, 3B:]^T (mlp.pf_bias_of)."""
    bd = mlp.bindings(mlp.skin_net_for(n_bones), "")[0]
    W1, b1 = P[bd.wname], P[bd.bname]
    aff = BoneAffine.apply(art[0], art[1], gauss)  # (M, 3B, 4)
    tab = torch.einsum("fc,mcj->mfj", W1[:, :3 * n_bones], aff)
    return torch.cat([tab[..., :3], tab[..., 3:] + (pf + b1)[..., None]], -1).contiguous()


class SkinChainA(Function):
    """The delta-skin MLP behind the per-frame table of skin_affine_table (LAB4D_NET_SKIN_A / _SKIN18_A, include/lab4d_mlp.h): points (S,3) ->
    relu(Wf[frame] [x; 1]) -> linear_2 -> linear_final.  Against SkinChain: no 96-wide MFMA layer, no stored bf16 copy of the coordinates and no
    weight-gradient launch for linear_1, no (S, 3B) coordinate gradient and no pass over it (k_bone_bwd_x + Gram): the backward chain kernel
    returns the point gradient (S,3) and dL/dWf (M,64,4) directly."""

    @staticmethod
    def forward(ctx, net, prec, spf, xyz, tab, *params):
        xyz = xyz.contiguous()
        need = any(ctx.needs_input_grad)
        inner = types.SimpleNamespace(needs_input_grad=(False, False, False, need, False, False, False, False, False) + tuple(ctx.needs_input_grad[5:]))
        out = mlp.MlpChain.forward(inner, net, prec, spf, xyz, None, None, -1, 0, None, *params, aff=tab.detach().contiguous())
        ctx.inner = inner
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out):
        res = mlp.MlpChain.backward(ctx.inner, d_out)
        g_tab = ctx.inner.g_aff
        ctx.inner = None
        return (None, None, None, res[3], g_tab) + tuple(res[9:])


# The delta-skin MLP runs in its per-frame affine form (SkinChainA); 0 restores the bone-coordinate form (SkinChain) for A/B measurements.
SKIN_AFFINE = os.environ.get("LAB4D_SKIN_AFFINE", "1") != "0"


class SkinBlend(Function):
    """skin weights + hemisphere-consistent dual-quaternion blend + apply (warping.py:322-333, geom_utils.py:45-83).
    The gaussian-scaled bone coordinates the skin weights depend on are recomputed inside the kernels from
    (art_r, art_d, gauss); their gradient is routed analytically (see csrc/skinning.hip k_blend_bwd)."""

    @staticmethod
    def forward(ctx, xyz, raw, art_r, art_d, gauss, se3_r, se3_d, spf):
        xyz, raw, art_r, art_d, gauss, se3_r, se3_d = [t.contiguous() for t in (xyz, raw, art_r, art_d, gauss, se3_r, se3_d)]
        _lib.require_device(xyz, raw, art_r, art_d, gauss, se3_r, se3_d)
        S, (M, B) = xyz.shape[0], se3_r.shape[:2]
        out = torch.empty(S, 3, device=xyz.device)
        ent = torch.empty(S, 1, device=xyz.device)
        dsk = torch.empty(S, 1, device=xyz.device)
        work = torch.empty(M * B * 12, device=xyz.device)
        with _lib.timed("k_blend_fwd", (0.0, 4.0 * S * (3 + B + 3 + 2))):  # xyz + raw read; out, entropy, delta_skin written
            _lib.check(_lib.lib().lab4d_skin_blend_forward(_lib.ptr(xyz), _lib.ptr(art_r), _lib.ptr(art_d), _lib.ptr(gauss), _lib.ptr(raw),
                                                           _lib.ptr(se3_r), _lib.ptr(se3_d), S, spf, M, B, _lib.ptr(out), _lib.ptr(ent), _lib.ptr(dsk),
                                                           _lib.ptr(work), _lib.stream()), "skin_blend_forward")
        ctx.save_for_backward(xyz, raw, art_r, art_d, gauss, se3_r, se3_d)
        ctx.spf = spf
        return out, ent, dsk

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out, g_ent, g_dsk):
        xyz, raw, art_r, art_d, gauss, se3_r, se3_d = ctx.saved_tensors
        S, (M, B) = xyz.shape[0], se3_r.shape[:2]
        g_out = g_out.contiguous()
        g_ent = g_ent.contiguous() if g_ent is not None else None
        g_dsk = g_dsk.contiguous() if g_dsk is not None else None
        gx, gr = torch.empty_like(xyz), torch.empty_like(raw)
        gse3 = torch.zeros(M, B, 8, device=xyz.device)
        need_p = any(ctx.needs_input_grad[2:5])
        gar = torch.empty_like(art_r) if need_p else None
        gad = torch.empty_like(art_d) if need_p else None
        gg = torch.zeros_like(gauss) if need_p else None
        work, fused = _blend_bwd_work(S, ctx.spf, M, B, xyz.device)  # fused: the per-frame reductions inside the kernel
        # one entry for the whole adjoint (k_blend_bwd incl. its two per-frame Gram reductions): reads xyz, raw, g_out, g_ent, g_dskin; writes
        # g_xyz, g_raw (unfused: also the (S, 2B+18) work arrays, which the Gram kernels read once more)
        with _lib.timed("k_blend_bwd+gram", (0.0, 4.0 * S * ((3 + B + 3 + 2) + (3 + B) + (0 if fused else 2 * (2 * B + 18))))):
            _lib.check(_lib.lib().lab4d_skin_blend_backward(_lib.ptr(xyz), _lib.ptr(art_r), _lib.ptr(art_d), _lib.ptr(gauss), _lib.ptr(raw),
                                                            _lib.ptr(se3_r), _lib.ptr(se3_d), _lib.ptr(g_out), _lib.ptr(g_ent), _lib.ptr(g_dsk), S,
                                                            ctx.spf, M, B, _lib.ptr(gx), _lib.ptr(gr), _lib.ptr(gse3), _lib.ptr(gar), _lib.ptr(gad),
                                                            _lib.ptr(gg), _lib.ptr(work), _lib.stream()), "skin_blend_backward")
        return gx, gr, gar, gad, gg, gse3[..., :4].contiguous(), gse3[..., 4:].contiguous(), None


class SkinBlendMulti(Function):
    """Several blends (SkinBlend) of the SAME points and delta-skin logits to different target transforms -- the two forward warps of a training
    query (skinning_warp_forward_multi).  Forward: SkinBlend's kernel once per target.  Backward: the adjoints of all targets land in ONE
    (S,3) / (S,B) pair (lab4d_skin_blend_backward_acc: the second and later launches add), where autograd would sum two (S,B) tensors in a
    pass of its own (5 GB of traffic per chunk at B = 25).  Arguments: xyz, raw, art_r, art_d, gauss, spf, then (se3_r, se3_d) per target;
    returns (out, entropy, delta_skin) per target, flattened."""

    @staticmethod
    def forward(ctx, xyz, raw, art_r, art_d, gauss, spf, *se3s):
        xyz, raw, art_r, art_d, gauss = [t.contiguous() for t in (xyz, raw, art_r, art_d, gauss)]
        se3s = [t.contiguous() for t in se3s]
        _lib.require_device(xyz, raw, art_r, art_d, gauss, *se3s)
        S, (M, B) = xyz.shape[0], art_r.shape[:2]
        outs = []
        for i in range(0, len(se3s), 2):
            out = torch.empty(S, 3, device=xyz.device)
            ent = torch.empty(S, 1, device=xyz.device)
            dsk = torch.empty(S, 1, device=xyz.device)
            work = torch.empty(M * B * 12, device=xyz.device)
            with _lib.timed("k_blend_fwd", (0.0, 4.0 * S * (3 + B + 3 + 2))):
                _lib.check(_lib.lib().lab4d_skin_blend_forward(_lib.ptr(xyz), _lib.ptr(art_r), _lib.ptr(art_d), _lib.ptr(gauss), _lib.ptr(raw),
                                                               _lib.ptr(se3s[i]), _lib.ptr(se3s[i + 1]), S, spf, M, B, _lib.ptr(out), _lib.ptr(ent),
                                                               _lib.ptr(dsk), _lib.ptr(work), _lib.stream()), "skin_blend_forward")
            outs += [out, ent, dsk]
        ctx.save_for_backward(xyz, raw, art_r, art_d, gauss, *se3s)
        ctx.spf = spf
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        xyz, raw, art_r, art_d, gauss, *se3s = ctx.saved_tensors
        S, (M, B) = xyz.shape[0], art_r.shape[:2]
        gx, gr = torch.empty_like(xyz), torch.empty_like(raw)
        need_p = any(ctx.needs_input_grad[2:5])
        # (the fused / unfused decision is a function of S -- with S = 0 both paths size their scratch M*B*34 -- so it is asked for the
        # real S, and that (work, fused) pair serves the first target instead of being thrown away)
        work0, fused = _blend_bwd_work(S, ctx.spf, M, B, xyz.device)
        gar = gad = gg = None
        gse3s = []
        first = True
        for i in range(0, len(se3s), 2):
            g_out, g_ent, g_dsk = grads[3 * (i // 2):3 * (i // 2) + 3]
            if g_out is None and g_ent is None and g_dsk is None:
                gse3s += [None, None]
                continue
            g_out = torch.zeros_like(xyz) if g_out is None else g_out.contiguous()
            g_ent = g_ent.contiguous() if g_ent is not None else None
            g_dsk = g_dsk.contiguous() if g_dsk is not None else None
            gse3 = torch.zeros(M, B, 8, device=xyz.device)
            par = torch.empty_like(art_r) if need_p else None
            pad = torch.empty_like(art_d) if need_p else None
            pg = torch.zeros_like(gauss) if need_p else None
            work = work0 if first else _blend_bwd_work(S, ctx.spf, M, B, xyz.device)[0]
            with _lib.timed("k_blend_bwd+gram", (0.0, 4.0 * S * ((3 + B + 3 + 2) + (3 + B) * (1 if first else 2) + (0 if fused else 2 * (2 * B + 18))))):
                _lib.check(_lib.lib().lab4d_skin_blend_backward_acc(_lib.ptr(xyz), _lib.ptr(art_r), _lib.ptr(art_d), _lib.ptr(gauss), _lib.ptr(raw),
                                                                    _lib.ptr(se3s[i]), _lib.ptr(se3s[i + 1]), _lib.ptr(g_out), _lib.ptr(g_ent),
                                                                    _lib.ptr(g_dsk), S, ctx.spf, M, B, _lib.ptr(gx), _lib.ptr(gr), _lib.ptr(gse3),
                                                                    _lib.ptr(par), _lib.ptr(pad), _lib.ptr(pg), _lib.ptr(work), 0 if first else 1,
                                                                    _lib.stream()), "skin_blend_backward_acc")
            first = False
            if need_p:  # (M,B)-sized: summed over the targets here
                gar, gad, gg = (par, pad, pg) if gar is None else (gar + par, gad + pad, gg + pg)
            gse3s += [gse3[..., :4].contiguous(), gse3[..., 4:].contiguous()]
        if first:
            return (None,) * (6 + len(se3s))
        return (gx, gr, gar, gad, gg, None) + tuple(gse3s)


def get_gauss(P):
    """SkinningField.get_gauss (skinning.py:142-153)."""
    lg = P["warp.skinning_model.log_gauss"]
    symm = P.get("warp.skinning_model.symm_idx")
    if symm is not None:
        lg = (lg[symm] + lg) / 2
    return lg.exp()


def _spf(shape):
    spf = 1
    for d in shape[1:-1]:
        spf *= d
    return spf


def skin_cond(t_embed, code, M):
    """Conditioning input of the delta-skin MLP's first layer: [time embedding | instance code] per frame."""
    return torch.cat([t_embed.expand(M, -1), code], -1)


def skin_logits(P, x, art, t_embed, code, M, spf, prec, pre=None):
    """The delta-skin field of SkinningField.forward (skinning.py:89-124) at the articulation `art`: gaussian-scaled bone
    coordinates -> delta-skin MLP.  x (S,3).  Returns the raw (S,B) MLP output and gauss (B,3).
    pre = {"gauss", "pf"}: the per-frame terms already evaluated by the step's prologue (deformable.frame_terms)."""
    gauss = pre["gauss"] if pre is not None else get_gauss(P)
    B = art[0].shape[1]
    net = mlp.skin_net_for(B)
    d, bd = mlp.describe(net), mlp.bindings(net, "")
    if SKIN_AFFINE and FUSE_BONE_COORDS:
        tab = pre.get("tab") if pre is not None else None
        if tab is None:
            pf = pre["pf"] if pre is not None else mlp.pf_bias_of(net, 0, P[bd[0].wname], skin_cond(t_embed, code, M))
            tab = skin_affine_table(P, art, gauss, pf, B)
        net_a = mlp.skin_net_for(B, affine=True)
        bda = mlp.bindings(net_a, "")
        params = []
        for l in range(mlp.describe(net_a).n_layers):
            params += [P[bda[l].wname], P[bda[l].bname]]
        return SkinChainA.apply(net_a, prec, spf, x, tab, *params), gauss
    pf = pre["pf"] if pre is not None else mlp.pf_bias_of(net, 0, P[bd[0].wname], skin_cond(t_embed, code, M))
    params = []
    for l in range(d.n_layers):
        params += [P[bd[l].wname], P[bd[l].bname]]
    if not FUSE_BONE_COORDS:
        bone = BoneCoords.apply(x, art[0], art[1], gauss, spf)  # (S,3B): input of the delta-skin MLP only
        return mlp.MlpChain.apply(net, prec, spf, bone, None, None, -1, 1, None, pf, *params), gauss
    return SkinChain.apply(net, prec, spf, x, art[0], art[1], gauss, 1, pf, *params), gauss


def skinning_warp(P, xyz, t_articulation, rest_articulation, t_embed, code, backward, prec=mlp.PREC_F32, pre=None):
    """SkinningWarp.forward (warping.py:277-336).  xyz: (M,N,D,3).  Returns warped xyz and
    {"skin_entropy","delta_skin"} (M,N,D,1).  t_embed: (M,128) per-frame (backward warp) or (1,128)
    mean embedding (forward warp, frame_id=None: warping.py:314).
    pre = {"se3", "gauss", "pf"}: per-frame terms of this warp evaluated by the step's prologue (deformable.frame_terms)."""
    shape = xyz.shape
    M, spf = shape[0], _spf(shape)
    if pre is not None:
        se3 = pre["se3"]
        art = t_articulation if backward else rest_articulation
    elif backward:
        se3 = Q.dual_quaternion_mul(rest_articulation, Q.dual_quaternion_inverse(t_articulation))
        art = t_articulation
    else:
        se3 = Q.dual_quaternion_mul(t_articulation, Q.dual_quaternion_inverse(rest_articulation))
        art = rest_articulation
    x = xyz.reshape(-1, 3)
    raw, gauss = skin_logits(P, x, art, t_embed, code, M, spf, prec, pre)
    out, ent, dsk = SkinBlend.apply(x, raw, art[0], art[1], gauss, se3[0], se3[1], spf)
    return out.view(shape), {"skin_entropy": ent.view(shape[:-1] + (1,)), "delta_skin": dsk.view(shape[:-1] + (1,))}


def skinning_warp_forward_multi(P, xyz, t_articulations, rest_articulation, t_embed_mean, code, prec=mlp.PREC_F32, pre=None):
    """Several FORWARD warps of the same canonical points to different target articulations (SkinningWarp.forward with
    backward=False, warping.py:306-333).  In the forward direction the skinning weights depend only on the points, the REST
    articulation, the mean time embedding and the instance code (warping.py:311-314: articulation = rest_articulation,
    frame_id = None) -- not on the target -- so the bone coordinates and the delta-skin MLP are evaluated ONCE and only the
    dual-quaternion blend runs per target.  The training graph warps every canonical sample forward twice (compute_flow into
    the pair partner's frame, nerf.py:966-973; cycle_loss into its own, deformable.py:173-198): the reference evaluates the
    skinning field twice with identical inputs, this evaluates it once.  Returns [(warped xyz, aux), ...].
    pre = {"gauss", "pf", "se3s": [(r, d) per target]}: the per-frame terms evaluated by the step's prologue (deformable.frame_terms)."""
    shape = xyz.shape
    M, spf = shape[0], _spf(shape)
    x = xyz.reshape(-1, 3)
    raw, gauss = skin_logits(P, x, rest_articulation, t_embed_mean, code, M, spf, prec, pre)
    rest_inv = None if pre is not None else Q.dual_quaternion_inverse(rest_articulation)
    se3s = [pre["se3s"][i] if pre is not None else Q.dual_quaternion_mul(t_art, rest_inv) for i, t_art in enumerate(t_articulations)]
    flat = SkinBlendMulti.apply(x, raw, rest_articulation[0], rest_articulation[1], gauss, spf, *[t for se3 in se3s for t in se3])
    outs = []
    for i in range(len(se3s)):
        out, ent, dsk = flat[3 * i:3 * i + 3]
        outs.append((out.view(shape), {"skin_entropy": ent.view(shape[:-1] + (1,)), "delta_skin": dsk.view(shape[:-1] + (1,))}))
    return outs


def dense_warp(P, xyz, t_embed, code, backward, prec=mlp.PREC_F32, prefix="warp.post_warp.", net=mlp.NET_DENSE):
    """DenseWarp.forward (warping.py:143-170): xyz + 0.1 * CondMLP([posenc6(xyz) | time embedding | instance code]) with the
    backward_map / forward_map weights.  t_embed: (M,128) output of the warp's own TimeEmbedding, code: (M,32).
    net: NET_DENSE = the D=2 post-warp of ComposedWarp (prefix "warp.post_warp."), NET_DENSE6 = fg_motion "dense" (prefix "warp.")."""
    shape = xyz.shape
    spf = 1
    for d in shape[1:-1]:
        spf *= d
    which = "backward_map." if backward else "forward_map."
    cond = torch.cat([t_embed, code], -1)
    conds = {0: cond, 4: cond} if net == mlp.NET_DENSE6 else {0: cond}  # the skip layer re-reads the whole input (base.py:74-75)
    motion = mlp.run_chain(net, prec, P, xyz.reshape(-1, 3), spf, conds=conds, prefix=prefix + which)
    return xyz + 0.1 * motion.view(shape)


def composed_warp(P, xyz, t_articulation, rest_articulation, t_embed, code, backward, prec=mlp.PREC_F32, dense=None):
    """ComposedWarp.forward (warping.py:445-483): skeleton skinning composed with a dense post-warp.  `dense` =
    {"t_embed": (M,128), "code_fw": (M,32), "code_bw": (M,32)} for a known frame_id, or None (frame_id is None: the post
    warp is skipped, warping.py:460,474)."""
    if not backward and dense is not None:
        xyz = dense_warp(P, xyz, dense["t_embed"], dense["code_fw"], False, prec)
    out, aux = skinning_warp(P, xyz, t_articulation, rest_articulation, t_embed, code, backward, prec)
    if backward and dense is not None:
        out = dense_warp(P, out, dense["t_embed"], dense["code_bw"], True, prec)
    return out, aux
