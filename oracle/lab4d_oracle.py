"""CPU oracle for the Lab4D differentiable-volume-rendering hot path.

TEST INFRASTRUCTURE ONLY -- never imported by the product path (lab4d_amd/).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it,
and only as the checker / baseline.

This is a plain PyTorch-CPU fp32 *restatement* of the reference algorithm
(lab4d-org/lab4d @ 2025-02-19); every function cites the reference file:line it
follows.  It is functional: weights live in a flat dict `P` whose keys are the
reference's own state_dict names (e.g. "basefield.linear_1.0.weight"), so a
reference checkpoint can be fed to it unchanged.  Per-frame quantities
(camera pose, bone articulations, appearance / time / instance codes) are
*inputs* -- they are produced by the reference's per-frame MLPs, which are
outside the hot path.

Parity pinning: tests/test_oracle_golden.py checks this file against golden
vectors produced by the reference's own code (tests/golden/make_golden.py, run
in the build container through oracle/ref_shim.py).  The reference's own tests
pin only PosEmbedding (tests/test_ops.py:64-133); everything else is pinned by
those generated goldens.

Shapes: M frames, N rays/frame, D samples/ray, B bones, payload last.
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# quaternion / dual quaternion algebra            (lab4d/utils/quat_transform.py)
# ----------------------------------------------------------------------------


def _pad_w(a):
    # quaternion.cu:46-57 -- a 3-vector operand is a pure quaternion with w = 0
    if a.shape[-1] == 3:
        a = torch.cat([torch.zeros_like(a[..., :1]), a], -1)
    return a


def quaternion_mul(a, b):
    """Hamilton product, real part first (quat_transform.py:62-81, quaternion.cu:29-64)."""
    a, b = torch.broadcast_tensors(_pad_w(a), _pad_w(b))
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack(
        (
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ),
        -1,
    )


def quaternion_conjugate(q):
    """quat_transform.py:27-43, quaternion.cu:202-217."""
    return torch.cat((q[..., :1], -q[..., 1:]), -1)


def quaternion_apply(q, p):
    """quat_transform.py:255-272:  (q * p) * conj(q), vector part."""
    return quaternion_mul(quaternion_mul(q, p), quaternion_conjugate(q))[..., 1:]


def quaternion_translation_apply(q, t, p):
    """quat_transform.py:275-279."""
    return quaternion_apply(q, p) + t


def quaternion_translation_inverse(q, t):
    """quat_transform.py:282-287."""
    qi = quaternion_conjugate(q)
    return qi, quaternion_apply(qi, -t)


def dual_quaternion_to_quaternion_translation(dq):
    """quat_transform.py:337-344:  t = 2 * (q_d * conj(q_r)).xyz"""
    qr, qd = dq
    t = 2 * quaternion_mul(qd, quaternion_conjugate(qr))[..., 1:]
    return qr, t


def dual_quaternion_mul(dq1, dq2):
    """quat_transform.py:430-438."""
    r1, d1 = dq1
    r2, d2 = dq2
    return quaternion_mul(r1, r2), quaternion_mul(r1, d2) + quaternion_mul(d1, r2)


def dual_quaternion_inverse(dq):
    """quat_transform.py:441-465: the quaternion conjugate of both parts."""
    return quaternion_conjugate(dq[0]), quaternion_conjugate(dq[1])


def dual_quaternion_apply(dq, p):
    """quat_transform.py:383-385."""
    q, t = dual_quaternion_to_quaternion_translation(dq)
    return quaternion_translation_apply(q, t, p)


def mat3x3_det(m):
    """matinv.cu:42-62; m: (...,3,3)."""
    x = m.reshape(m.shape[:-2] + (9,))
    return (
        x[..., 0] * x[..., 4] * x[..., 8]
        + x[..., 3] * x[..., 7] * x[..., 2]
        + x[..., 6] * x[..., 5] * x[..., 1]
        - x[..., 2] * x[..., 4] * x[..., 6]
        - x[..., 5] * x[..., 7] * x[..., 0]
        - x[..., 8] * x[..., 3] * x[..., 1]
    )


def mat3x3_scale_adjoint(m, scales):
    """matinv.cu:80-116: adjugate(m) / scales."""
    x = m.reshape(m.shape[:-2] + (9,))
    s = (1.0 / scales)[..., None]
    adj = torch.stack(
        (
            x[..., 4] * x[..., 8] - x[..., 5] * x[..., 7],
            x[..., 2] * x[..., 7] - x[..., 1] * x[..., 8],
            x[..., 1] * x[..., 5] - x[..., 2] * x[..., 4],
            x[..., 5] * x[..., 6] - x[..., 3] * x[..., 8],
            x[..., 0] * x[..., 8] - x[..., 2] * x[..., 6],
            x[..., 2] * x[..., 3] - x[..., 0] * x[..., 5],
            x[..., 3] * x[..., 7] - x[..., 4] * x[..., 6],
            x[..., 1] * x[..., 6] - x[..., 0] * x[..., 7],
            x[..., 0] * x[..., 4] - x[..., 1] * x[..., 3],
        ),
        -1,
    )
    return (s * adj).reshape(m.shape)


def mat3x3_inv(m):
    """mat3x3.py:66-97 / matinv.cu mat3x3_inv_forward: adjugate / determinant."""
    return mat3x3_scale_adjoint(m, mat3x3_det(m))


# ----------------------------------------------------------------------------
# encodings and MLP stacks           (lab4d/nnutils/embedding.py, base.py)
# ----------------------------------------------------------------------------


def pos_embedding(x, n_freqs, alpha=None):
    """PosEmbedding.forward (embedding.py:69-125).

    out = [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), cos(2^1 x), ...]; channel order is
    (freq, {sin,cos}, in_channel).  `alpha` is the coarse-to-fine annealing window.
    """
    if n_freqs == -1:
        return x[..., :0]
    if n_freqs == 0:
        return x
    c = x.shape[-1]
    freq = 2.0 ** torch.arange(n_freqs, dtype=x.dtype)  # 2**linspace(0, L-1, L)
    ang = freq[:, None] * x[..., None, :]  # (..., L, C)
    bands = torch.stack([torch.sin(ang), torch.cos(ang)], -2)  # (..., L, 2, C)
    if alpha is not None:
        w = torch.clamp(alpha * n_freqs - torch.arange(n_freqs, dtype=x.dtype), 0.0, 1.0)
        w = 0.5 * (1 + torch.cos(math.pi * w + math.pi))
        bands = bands * w[:, None, None]
    return torch.cat([x, bands.reshape(x.shape[:-1] + (2 * n_freqs * c,))], -1)


def base_mlp(P, prefix, x, D, skips=(4,), final_act=False):
    """BaseMLP.forward (base.py:65-78): D x (Linear+ReLU), skip = cat([x, out]) *input first*,
    then linear_final (+ReLU if final_act)."""
    out = x
    for i in range(D):
        if i in skips:
            out = torch.cat([x, out], -1)
        out = F.relu(F.linear(out, P[f"{prefix}.linear_{i+1}.0.weight"], P[f"{prefix}.linear_{i+1}.0.bias"]))
    if final_act:
        out = F.relu(F.linear(out, P[f"{prefix}.linear_final.0.weight"], P[f"{prefix}.linear_final.0.bias"]))
    else:
        out = F.linear(out, P[f"{prefix}.linear_final.weight"], P[f"{prefix}.linear_final.bias"])
    return out


def inst_code(P, prefix, inst_id):
    """InstEmbedding.forward (embedding.py:246-264), beta_prob == 0 (no code swapping).
    num_inst == 1 -> mapping(zeros_like(inst_id))."""
    w = P[f"{prefix}.inst_embedding.mapping.weight"]
    if w.shape[0] == 1:
        inst_id = torch.zeros_like(inst_id)
    return w[inst_id]


def cond_mlp(P, prefix, feat, code, D, skips=(4,), final_act=False):
    """CondMLP.forward (base.py:123-150): the per-frame instance code (M,C) is broadcast over
    the sample axes and appended *after* the features."""
    code = code.view(code.shape[:1] + (1,) * (feat.ndim - 2) + (-1,)).expand(feat.shape[:-1] + (-1,))
    return base_mlp(P, prefix, torch.cat([feat, code], -1), D, skips, final_act)


# ----------------------------------------------------------------------------
# ray sampling and compositing                (lab4d/utils/render_utils.py)
# ----------------------------------------------------------------------------


def sample_cam_rays(hxy, Kinv, near_far, n_depth=64, depth=None):
    """render_utils.py:8-56 with perturb=False (the only mode any caller uses)."""
    M, N = hxy.shape[:2]
    dirs = torch.einsum("mni,mij->mnj", hxy, Kinv.permute(0, 2, 1))
    dnorm = torch.norm(dirs, dim=-1)
    if depth is None:
        z = torch.linspace(0, 1, n_depth)[None]
        depth = near_far[:, 0:1] * (1 - z) + near_far[:, 1:2] * z
        depth = depth[:, None, :, None].repeat(1, N, 1, 1)
    else:
        n_depth = depth.shape[2]
    xyz = dirs.unsqueeze(2) * depth
    deltas = depth[:, :, 1:] - depth[:, :, :-1]
    deltas = torch.cat([deltas, deltas[:, :, -1:]], -2) * dnorm[..., None, None]
    dirs = (dirs / dnorm.unsqueeze(-1)).unsqueeze(2).repeat(1, 1, n_depth, 1)
    return xyz, dirs, deltas, depth


def compute_weights(density, deltas):
    """render_utils.py:99-126.  tau = sigma*delta; w_i = (1-exp(-tau_i)) * exp(-sum_{j<i} tau_j);
    returns (weights, transmit) with transmit_i = exp(-sum_{j<=i} tau_j)."""
    tau = (deltas * density)[..., 0]
    alpha = 1 - torch.exp(-tau)
    T = torch.exp(-torch.cumsum(tau, -1))
    T_excl = torch.cat([torch.ones_like(T[..., :1]), T[..., :-1]], -1)
    return alpha * T_excl, T


def integrate(field_dict, weights):
    """render_utils.py:129-184."""
    key_skip = ["density", "vis", "flow", "eikonal", "xy_reproj", "xyz_reproj", "gauss_density"]
    key_freeze = ["cyc_dist", "xyz_cam", "skin_entropy"]
    out = {"mask": weights.sum(-1, keepdim=True)}
    wn = weights / (out["mask"] + 1e-6)
    for k, v in field_dict.items():
        if k in key_skip:
            continue
        wt = wn.detach() if k in key_freeze else wn
        out[k] = (wt.unsqueeze(-1) * v).sum(-2)
    if "flow" in field_dict:
        wf = weights * field_dict["flow"][..., 2]
        wf = wf / (wf.sum(-1, keepdim=True) + 1e-6)
        out["flow"] = (wf.unsqueeze(-1) * field_dict["flow"][..., :2]).sum(-2)
    if "normal" in field_dict:
        out["normal"] = F.normalize(out["normal"], 2, -1)
    dkeys = [k for k in out if "density_" in k]
    dsum = torch.cat([out[k] for k in dkeys], -1).sum(-1, keepdim=True) + 1e-6
    for k in dkeys:
        out[k.replace("density_", "mask_")] = out[k] / dsum
        del out[k]
    return out


def render_pixel(field_dict, deltas):
    """render_utils.py:59-96."""
    weights, transmit = compute_weights(field_dict["density"], deltas)
    out = integrate(field_dict, weights)
    if "eikonal" in field_dict:
        out["eikonal"] = field_dict["eikonal"].mean(dim=(-1, -2))
    if "delta_skin" in field_dict:
        out["delta_skin"] = field_dict["delta_skin"].mean(dim=(-1, -2))
    vis_t = transmit[..., None].detach()
    vis_loss = -(F.logsigmoid(field_dict["vis"]) * vis_t).mean(-2)
    out["vis"] = vis_loss / vis_t.mean().detach()
    if "gauss_density" in field_dict:
        gw, _ = compute_weights(field_dict["gauss_density"], deltas)
        out["gauss_mask"] = gw.sum(-1, keepdim=True)
    return out


def sample_pdf(bins, weights, n_importance, eps=1e-5, return_inds=False):
    """render_utils.py:187-233 with det=True (the only mode used: nerf.py:721-727).
    `inds` (int64) must be reproduced bit-exactly by the device path."""
    n_rays, n_samples = weights.shape
    weights = weights + eps
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = torch.linspace(0, 1, n_importance).expand(n_rays, n_importance).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, n_samples)
    g = torch.stack([below, above], -1).view(n_rays, 2 * n_importance)
    cdf_g = torch.gather(cdf, 1, g).view(n_rays, n_importance, 2)
    bins_g = torch.gather(bins, 1, g).view(n_rays, n_importance, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    samples = bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])
    if return_inds:
        return samples, inds
    return samples


# ----------------------------------------------------------------------------
# rigid transforms between camera and field          (lab4d/nnutils/nerf.py)
# ----------------------------------------------------------------------------


def _expand_se3(se3, shape):
    q, t = se3
    q = q.view(q.shape[:1] + (1,) * (len(shape) - 2) + (4,)).expand(shape[:-1] + (4,))
    t = t.view(t.shape[:1] + (1,) * (len(shape) - 2) + (3,)).expand(shape[:-1] + (3,))
    return q, t


def cam_to_field(xyz_cam, dir_cam, field2cam):
    """NeRF.cam_to_field (nerf.py:821-844)."""
    qi, ti = quaternion_translation_inverse(field2cam[0], field2cam[1])
    q, t = _expand_se3((qi, ti), xyz_cam.shape)
    xyz = quaternion_translation_apply(q, t, xyz_cam)
    dirs = quaternion_apply(q, dir_cam) if dir_cam is not None else None
    return xyz, dirs


def field_to_cam(xyz, field2cam):
    """NeRF.field_to_cam (nerf.py:846-863)."""
    q, t = _expand_se3(field2cam, xyz.shape)
    return quaternion_translation_apply(q, t, xyz)


def kmatinv(Kinv):
    """geom_utils.Kmatinv (geom_utils.py:322-341): invert an upper-triangular pinhole matrix
    [[fx,0,px],[0,fy,py],[0,0,1]] analytically (acts on both K and K^-1)."""
    fx, fy = Kinv[..., 0, 0], Kinv[..., 1, 1]
    px, py = Kinv[..., 0, 2], Kinv[..., 1, 2]
    out = torch.zeros_like(Kinv)
    out[..., 0, 0] = 1.0 / fx
    out[..., 1, 1] = 1.0 / fy
    out[..., 0, 2] = -px / fx
    out[..., 1, 2] = -py / fy
    out[..., 2, 2] = 1
    return out


def pinhole_projection(Kmat, xyz_cam):
    """geom_utils.py:14-27."""
    shape = xyz_cam.shape
    K = Kmat.view(shape[:1] + (1,) * (len(shape) - 2) + (3, 3))
    hxy = torch.einsum("...ij,...j->...i", K, xyz_cam)
    return hxy / (hxy[..., -1:] + 1e-6)


# ----------------------------------------------------------------------------
# linear-blend skinning with dual quaternions
#   (nnutils/warping.py:277-336, skinning.py:89-153, utils/transforms.py:9-40,
#    utils/geom_utils.py:45-83, utils/loss_utils.py:21-42)
# ----------------------------------------------------------------------------


def get_gauss(P):
    """SkinningField.get_gauss (skinning.py:142-153) -- symm_idx handled by the caller via
    P["warp.skinning_model.symm_idx"] (absent -> None)."""
    lg = P["warp.skinning_model.log_gauss"]
    symm = P.get("warp.skinning_model.symm_idx", None)
    if symm is not None:
        lg = (lg[symm] + lg) / 2
    return lg.exp()


def get_bone_coords(xyz, bone2obj):
    """transforms.py:9-25: apply the inverse bone transform of every bone to every point."""
    obj2bone = dual_quaternion_inverse(bone2obj)
    B = bone2obj[0].shape[-2]
    xyz = xyz[..., None, :].expand(xyz.shape[:-1] + (B, 3))
    return dual_quaternion_apply(obj2bone, xyz)


def skinning_field(P, xyz, bone2obj, t_embed, code):
    """SkinningField.forward (skinning.py:89-124).
    bone2obj: ((M,B,4),(M,B,4)); t_embed: (M,128) or (1,128) (mean embedding for forward warps);
    code: (M,32) delta_field instance code.  Returns skin (M,N,D,B), delta (M,N,D,B)."""
    M = xyz.shape[0]
    b2o = tuple(a[:, None, None].expand(xyz.shape[:3] + a.shape[1:]) for a in bone2obj)
    xyz_bone = get_bone_coords(xyz, b2o) / get_gauss(P).view(1, 1, 1, -1, 3)
    dist2 = xyz_bone.pow(2).sum(-1)
    emb = xyz_bone.reshape(xyz.shape[:-1] + (-1,))  # PosEmbedding(3B, 0) is the identity
    t = t_embed.reshape(-1, 1, 1, t_embed.shape[-1]).expand(xyz.shape[:-1] + (-1,))
    delta = cond_mlp(P, "warp.skinning_model.delta_field", torch.cat([emb, t], -1), code, D=2, final_act=False)
    delta = F.relu(delta) * 0.1
    return -(dist2 + delta), delta


def dual_quaternion_skinning(dq, pts, skin):
    """geom_utils.py:45-83: hemisphere-consistent weighted blend of bone dual quaternions."""
    shape = pts.shape
    M, B, _ = dq[0].shape
    pts = pts.reshape(M, -1, 3)
    skin = skin.reshape(M, -1, B)
    n = pts.shape[1]
    qr = dq[0][:, None].expand(M, n, B, 4)
    qd = dq[1][:, None].expand(M, n, B, 4)
    anchor = skin.argmax(-1).view(M, n, 1, 1).expand(M, n, 1, 4)
    sign = ((torch.gather(qr, 2, anchor) * qr).sum(-1) > 0)[..., None].to(pts.dtype) * 2 - 1
    qr = sign * qr
    qd = sign * qd
    qr_w = torch.einsum("bnk,bnkl->bnl", skin, qr)
    qd_w = torch.einsum("bnk,bnkl->bnl", skin, qd)
    inv = qr_w.norm(p=2, dim=-1, keepdim=True).reciprocal()
    out = dual_quaternion_apply((qr_w * inv, qd_w * inv), pts)
    return out.view(shape)


def cross_entropy_skin_loss(skin):
    """loss_utils.py:21-42: CE of the logits against their own arg-max one-hot
    = logsumexp(skin) - max(skin)."""
    return torch.logsumexp(skin, -1) - skin.max(-1)[0]


def skinning_warp(P, xyz, t_articulation, rest_articulation, t_embed, code, backward):
    """SkinningWarp.forward (warping.py:277-336).
    backward=True : time-t -> canonical; bones evaluated at t_articulation, t_embed = per-frame.
    backward=False: canonical -> time-t; bones at rest_articulation, t_embed = mean embedding
                    (frame_id=None, warping.py:314).
    Returns warped xyz and aux {"skin_entropy","delta_skin"} (M,N,D,1)."""
    if backward:
        se3 = dual_quaternion_mul(rest_articulation, dual_quaternion_inverse(t_articulation))
        art = t_articulation
    else:
        se3 = dual_quaternion_mul(t_articulation, dual_quaternion_inverse(rest_articulation))
        art = rest_articulation
    skin, delta = skinning_field(P, xyz, art, t_embed, code)
    out = dual_quaternion_skinning(se3, xyz, skin.softmax(-1))
    aux = {
        "skin_entropy": cross_entropy_skin_loss(skin)[..., None],
        "delta_skin": delta.pow(2).mean(-1, keepdim=True),
    }
    return out, aux


def gauss_density(P, xyz, rest_articulation):
    """Deformable.compute_gauss_density + SkinningWarp.get_gauss_density
    (deformable.py:329-356, warping.py:355-387, transforms.py:28-40).  Uses frame 0's rest bones."""
    shape = xyz.shape[:-1]
    b2o = (rest_articulation[0][:1], rest_articulation[1][:1])
    _, center = dual_quaternion_to_quaternion_translation(b2o)  # (1,B,3)
    pts = xyz.reshape(-1, 3)
    dist2 = (pts[:, None, :] - center).pow(2).sum(-1) / (0.01**2)
    dens = (-0.5 * dist2).exp().max(-1)[0][..., None]
    return (dens * P["warp.logibeta"].exp()).view(shape + (1,))


# ----------------------------------------------------------------------------
# the fields                         (nnutils/nerf.py, visibility.py, feature.py)
# ----------------------------------------------------------------------------


def nerf_forward(P, xyz, codes, with_color=True, appr_code=None, get_density=True, alpha=None, cfg=None, dir=None):
    """NeRF.forward (nerf.py:167-215), fg configuration: num_freq_dir=-1 (no view dependence),
    appearance code appended to the rgb head input.

    codes: {"basefield": (M,32), "colorfield": (M,32)} instance codes; appr_code: (M,32)."""
    cfg = cfg or FG_CFG
    feat = cond_mlp(P, "basefield", pos_embedding(xyz, cfg["num_freq_xyz"], alpha), codes["basefield"],
                    D=cfg["D"], final_act=True)
    sdf = F.linear(feat, P["sdf.weight"], P["sdf.bias"])
    if get_density:
        ibeta = P["logibeta"].exp()
        out = (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() * ibeta)) * ibeta  # VolSDF Laplace CDF
    else:
        out = sdf
    if not with_color:
        return out
    cfeat = cond_mlp(P, "colorfield", pos_embedding(xyz, cfg["num_freq_xyz"] + 2, alpha), codes["colorfield"],
                     D=2, final_act=True)
    feat = feat + cfeat
    if dir is not None:  # view direction embedding (nerf.py:196,211); bg field: num_freq_dir = 0 -> the raw 3-vector
        feat = torch.cat([feat, pos_embedding(dir, cfg.get("num_freq_dir", 0))], -1)
    if appr_code is not None:
        a = appr_code.view(appr_code.shape[:1] + (1,) * (xyz.ndim - 2) + (-1,)).expand(xyz.shape[:-1] + (-1,))
        feat = torch.cat([feat, a], -1)
    h = F.relu(F.linear(feat, P["rgb.0.weight"], P["rgb.0.bias"]))
    rgb = torch.sigmoid(F.linear(h, P["rgb.2.weight"], P["rgb.2.bias"]))
    return rgb, out


FG_CFG = {"D": 8, "W": 256, "num_freq_xyz": 10}
BG_CFG = {"D": 5, "W": 128, "num_freq_xyz": 6, "num_freq_dir": 0}


def vis_field(P, xyz, code):
    """VisField.forward (visibility.py:53-63)."""
    return cond_mlp(P, "vis_mlp.basefield", pos_embedding(xyz, 10), code, D=2, final_act=False)


def dense_warp(P, xyz, t_embed, code, backward, prefix="warp.post_warp", D=2):
    """DenseWarp.forward (warping.py:143-170): xyz + 0.1 * CondMLP(D)([posenc6 | time embedding | instance code]); D = 2 for ComposedWarp's
    post-warp (warping.py:432-434), 6 (the class default, skip connection at layer 4) for fg_motion "dense"."""
    m = f"{prefix}.backward_map" if backward else f"{prefix}.forward_map"
    te = t_embed.view(t_embed.shape[:1] + (1,) * (xyz.ndim - 2) + (-1,)).expand(xyz.shape[:-1] + (-1,))
    feat = torch.cat([pos_embedding(xyz, 6), te], -1)
    return xyz + 0.1 * cond_mlp(P, m, feat, code, D=D, final_act=False)


def composed_warp(P, xyz, t_articulation, rest_articulation, t_embed, code, backward, dense=None):
    """ComposedWarp.forward (warping.py:445-483)."""
    if not backward and dense is not None:
        xyz = dense_warp(P, xyz, dense["t_embed"], dense["code_fw"], False)
    out, aux = skinning_warp(P, xyz, t_articulation, rest_articulation, t_embed, code, backward)
    if backward and dense is not None:
        out = dense_warp(P, out, dense["t_embed"], dense["code_bw"], True)
    return out, aux


def compute_feat(P, xyz):
    """FeatureNeRF.compute_feat (feature.py:136-150)."""
    f = base_mlp(P, "feature_field", pos_embedding(xyz, 6), D=5, final_act=False)
    return f / f.norm(dim=-1, keepdim=True)


def global_match(P, feat_px, feat_canonical, xyz_canonical, perm):
    """FeatureNeRF.global_match (feature.py:152-199); `perm` = the host-drawn randperm[:1024]."""
    shape = feat_px.shape
    fp = feat_px.reshape(-1, shape[-1])
    fc = feat_canonical.reshape(-1, shape[-1])[perm]
    xc = xyz_canonical.reshape(-1, 3)[perm]
    prob = torch.softmax(fp @ fc.t() * P["logsigma"].exp(), 1)
    return (prob.unsqueeze(-1) * xc).sum(1).view(shape[:-1] + (-1,))


def eikonal_sdf_grad(P, xyz, code, alpha=None, create_graph=True):
    """torch_utils.compute_gradient (torch_utils.py:4-27) of sdf wrt xyz."""
    with torch.enable_grad():
        x = xyz.detach().requires_grad_(True)
        sdf = nerf_forward(P, x, {"basefield": code}, with_color=False, get_density=False, alpha=alpha)
        (g,) = torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=create_graph)
    return g


def compute_eikonal(P, xyz, code, rand_inds, alpha=None):
    """NeRF.compute_eikonal (nerf.py:416-453).  rand_inds: host-drawn multinomial ray subset
    (indices into M*N) or None for all rays."""
    M, N, D, _ = xyz.shape
    pts = xyz.reshape(-1, D, 3)
    c = code[:, None].expand(M, N, code.shape[-1]).reshape(M * N, -1)
    out = torch.zeros_like(pts[..., 0])
    if rand_inds is None:
        rand_inds = torch.arange(M * N)
    g = eikonal_sdf_grad(P, pts[rand_inds].detach(), c[rand_inds], alpha)
    out[rand_inds] = (g.norm(2, dim=-1) - 1) ** 2
    return out.reshape(M, N, D, 1)


# ----------------------------------------------------------------------------
# Deformable.query_field, training graph       (nerf.py:580-684, feature.py:89-134,
#                                               deformable.py:300-327)
# ----------------------------------------------------------------------------


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


def query_field_train(P, fr, hxy, rng, flow_thresh=None, n_depth=64, alpha=None):
    """Training-mode Deformable.query_field for a SkinningWarp ("bob"/"skel-*") foreground.

    fr: per-frame inputs {Kinv, field2cam=(q,t), near_far, t_articulation, rest_articulation,
        t_embed (M,128), t_embed_mean (1,128), appr_code (M,32), code_base/code_color/code_vis/
        code_skin (M,32), feature (M,N,16)}
    rng: host-drawn randomness {"eik_inds": LongTensor|None, "match_perm": LongTensor}
    fr["dense"] (optional): {"t_embed" (M,128), "code_fw" (M,32), "code_bw" (M,32)} = the per-frame inputs of ComposedWarp's
        dense post-warp (fg_motion "comp_skel-*_dense", warping.py:445-483); every warp below then goes through
        composed_warp.  The flow branch warps into the pair partner's frame, so its post-warp sees the partner's time
        embedding (nerf.py:966-973: frame_id_next).
    fr["motion"] (optional): "rigid" = fg_motion "rigid", the reference's default (warping.py:35-36,59-91: IdentityWarp -- every warp is the
        identity, the skin terms stay the zeros of NeRF.cycle_loss, nerf.py:905-927, and there is no gaussian-bone density, deformable.py:344);
        "dense" = fg_motion "dense" (warping.py:37-38,94-170: a bare DenseWarp(D=6, W=256) with parameters under "warp.").
    Returns feat_dict, deltas, aux_dict exactly as the reference does."""
    dense = fr.get("dense")
    motion = fr.get("motion", "skinning")

    def warp(x, t_art, rest_art, t_embed, backward, partner=False):
        if motion == "rigid":
            return x, {}
        if motion == "dense":
            te = flip_pair(fr["t_embed_dense"]) if partner else fr["t_embed_dense"]
            return dense_warp(P, x, te, fr["code_dense_bw" if backward else "code_dense_fw"], backward, prefix="warp", D=6), {}
        if dense is None:
            return skinning_warp(P, x, t_art, rest_art, t_embed, fr["code_skin"], backward=backward)
        d = dict(dense, t_embed=flip_pair(dense["t_embed"])) if partner else dense
        return composed_warp(P, x, t_art, rest_art, t_embed, fr["code_skin"], backward, dense=d)

    xyz_cam, dir_cam, deltas, depth = sample_cam_rays(hxy, fr["Kinv"], fr["near_far"], n_depth=n_depth)
    # backward warp (deformable.py:119-152)
    xyz_t, _ = cam_to_field(xyz_cam, None, fr["field2cam"])
    xyz, bw_aux = warp(xyz_t, fr["t_articulation"], fr["rest_articulation"], fr["t_embed"], True)
    fd = {}
    vis = vis_field(P, xyz, fr["code_vis"])
    rgb, density = nerf_forward(P, xyz, {"basefield": fr["code_base"], "colorfield": fr["code_color"]},
                                appr_code=fr["appr_code"], alpha=alpha)
    fd["rgb"], fd["density"], fd["density_fg"] = rgb, density, density
    fd["vis"] = vis
    # flow (nerf.py:948-997): warp canonical points into the pair partner's camera
    nxt = flip_pair({k: fr[k] for k in ["Kinv", "field2cam", "t_articulation", "rest_articulation"]})
    xyz_next, _ = warp(xyz, nxt["t_articulation"], nxt["rest_articulation"], fr["t_embed_mean"], False, partner=True)
    xyz_cam_next = field_to_cam(xyz_next, nxt["field2cam"])
    hxy_next = pinhole_projection(kmatinv(nxt["Kinv"]), xyz_cam_next)
    flow = (hxy_next - hxy.unsqueeze(-2))[..., :2]
    valid = xyz_cam_next[..., -1:] > 1e-6
    if flow_thresh is not None:
        valid = valid & (flow.norm(dim=-1, keepdim=True) < float(flow_thresh))
    fd["flow"] = torch.cat([flow, valid.to(flow.dtype)], -1)
    # cycle loss (deformable.py:173-198)
    xyz_cyc, cyc_aux = warp(xyz, fr["t_articulation"], fr["rest_articulation"], fr["t_embed_mean"], False)
    fd["cyc_dist"] = (xyz_cyc - xyz_t).norm(2, -1, keepdim=True)
    for k in ["skin_entropy", "delta_skin"]:
        # NeRF.cycle_loss's zeros (nerf.py:905-927) unless the warp reports the term (nerf.py:658-664)
        fd[k] = (cyc_aux[k] + bw_aux[k]) / 2 if k in cyc_aux else torch.zeros_like(fd["cyc_dist"])
    # eikonal on a ray subset (nerf.py:740-767)
    fd["eikonal"] = compute_eikonal(P, xyz, fr["code_base"], rng.get("eik_inds"), alpha)
    fd["xyz"] = xyz
    fd["xyz_cam"] = xyz_cam
    fd["depth"] = depth / P["logscale"].exp()
    # feature field + matching (feature.py:89-134)
    fd["feature"] = compute_feat(P, xyz)
    aux = {}
    xyz_matches = global_match(P, fr["feature"], fd["feature"], xyz, rng["match_perm"])
    xm_next, _ = warp(xyz_matches[:, :, None], fr["t_articulation"], fr["rest_articulation"], fr["t_embed_mean"], False)
    xyz_reproj = field_to_cam(xm_next, fr["field2cam"])[:, :, 0]
    aux["xyz_matches"] = xyz_matches
    aux["xyz_reproj"] = xyz_reproj
    aux["xy_reproj"] = pinhole_projection(kmatinv(fr["Kinv"]), xyz_reproj)[..., :2]
    if motion not in ("rigid", "dense"):  # deformable.py:344: SkinningWarp (and its ComposedWarp subclass) only
        fd["gauss_density"] = gauss_density(P, xyz, fr["rest_articulation"])
    return fd, deltas, aux


def render_train(P, fr, hxy, rng, flow_thresh=None, n_depth=64, alpha=None):
    """dvr_model.render_samples for field_type == "fg" (model.py:328-361): one field, so
    compose_fields is the identity and `rendered` == aux_dict["fg"] (+ matches)."""
    fd, deltas, aux = query_field_train(P, fr, hxy, rng, flow_thresh, n_depth, alpha)
    rendered = render_pixel(fd, deltas)
    aux_fg = dict(aux)
    aux_fg.update(render_pixel(fd, deltas))
    rendered["xyz_matches"] = aux["xyz_matches"]
    rendered["xyz_reproj"] = aux["xyz_reproj"]
    return {"rendered": rendered, "aux_dict": {"fg": aux_fg}}


# ----------------------------------------------------------------------------
# evaluation graph: importance sampling, valid-mask compaction, normals
#                                       (nerf.py:455-528, 605-636, 686-738, 769-819)
# ----------------------------------------------------------------------------


def extend_aabb(aabb, factor=0.1):
    """geom_utils.py:409-422."""
    ext = (aabb[1] - aabb[0]) * factor
    return torch.stack([aabb[0] - ext, aabb[1] + ext], 0)


def check_inside_aabb(xyz, aabb):
    """geom_utils.py:506-517 (strict inequalities)."""
    return ((xyz > aabb[:1]) & (xyz < aabb[1:])).all(-1)


def backward_warp_of(P, fr, xyz_t):
    """Deformable.backward_warp's warp step (deformable.py:139-146) for every fg_motion restated here (see query_field_train)."""
    motion = fr.get("motion", "skinning")
    if motion == "rigid":
        return xyz_t
    if motion == "dense":
        return dense_warp(P, xyz_t, fr["t_embed_dense"], fr["code_dense_bw"], True, prefix="warp", D=6)
    if fr.get("dense") is not None:
        return composed_warp(P, xyz_t, fr["t_articulation"], fr["rest_articulation"], fr["t_embed"], fr["code_skin"], True, dense=fr["dense"])[0]
    return skinning_warp(P, xyz_t, fr["t_articulation"], fr["rest_articulation"], fr["t_embed"], fr["code_skin"], backward=True)[0]


def get_valid_idx(P, xyz, xyz_t, t_articulation):
    """NeRF.get_valid_idx (nerf.py:495-528): inside extend_aabb(aabb,0.1) AND (when the samples carry articulations: SkinningWarp
    fields, deformable.py:254-289) inside the frame-0 bone-centre aabb extended by 1.0.  bool (M,N,D); must be bit-exact."""
    valid = check_inside_aabb(xyz, extend_aabb(P["aabb"]))
    if t_articulation is None:
        return valid
    _, tb = dual_quaternion_to_quaternion_translation(t_articulation)
    tb = tb[0]
    t_aabb = extend_aabb(torch.stack([tb.min(0)[0], tb.max(0)[0]], 0), factor=1.0)
    return valid & check_inside_aabb(xyz_t, t_aabb)


def importance_sampling(P, fr, hxy, n_depth=64, alpha=None):
    """NeRF.importance_sampling (nerf.py:686-738): n/2 coarse + n/2 inverse-CDF samples, sorted."""
    nc = n_depth // 2
    xyz_cam, _, deltas, depth = sample_cam_rays(hxy, fr["Kinv"], fr["near_far"], n_depth=nc)
    xyz_t, _ = cam_to_field(xyz_cam, None, fr["field2cam"])
    xyz = backward_warp_of(P, fr, xyz_t)
    density = nerf_forward(P, xyz, {"basefield": fr["code_base"]}, with_color=False, alpha=alpha)
    weights, _ = compute_weights(density, deltas)
    depth_mid = (0.5 * (depth[:, :, :-1] + depth[:, :, 1:])).view(-1, nc - 1)
    new, inds = sample_pdf(depth_mid, weights.view(-1, nc)[:, 1:-1], nc, return_inds=True)
    depth_all, _ = torch.sort(torch.cat([depth, new.reshape(depth.shape)], -2), -2)
    return sample_cam_rays(hxy, fr["Kinv"], fr["near_far"], depth=depth_all), inds


def query_field_eval(P, fr, hxy, n_depth=64, alpha=None):
    """Eval-mode Deformable.query_field (train-only fields return {}, decorator.py:4-17)."""
    (xyz_cam, dir_cam, deltas, depth), inds = importance_sampling(P, fr, hxy, n_depth, alpha)
    xyz_t, _ = cam_to_field(xyz_cam, None, fr["field2cam"])
    xyz = backward_warp_of(P, fr, xyz_t)
    vis = vis_field(P, xyz, fr["code_vis"])
    has_bones = fr.get("motion", "skinning") not in ("rigid", "dense")
    valid = get_valid_idx(P, xyz, xyz_t, fr["t_articulation"] if has_bones else None)
    # query_nerf with compaction (nerf.py:782-819): invalid samples get rgb = density = 0
    rgb, density = nerf_forward(P, xyz, {"basefield": fr["code_base"], "colorfield": fr["code_color"]},
                                appr_code=fr["appr_code"], alpha=alpha)
    rgb = rgb * valid[..., None]
    density = density * valid[..., None]
    fd = {"rgb": rgb, "density": density, "density_fg": density, "vis": vis}

    # normals in camera space through the whole warp (nerf.py:455-493)
    def fn_sdf(xc):
        xt, _ = cam_to_field(xc, None, fr["field2cam"])
        xx = backward_warp_of(P, fr, xt)
        return nerf_forward(P, xx, {"basefield": fr["code_base"]}, with_color=False, get_density=False, alpha=alpha)

    with torch.enable_grad():
        xc = xyz_cam.detach().requires_grad_(True)
        s = fn_sdf(xc)
        (g,) = torch.autograd.grad(s, xc, torch.ones_like(s))
    fd["eikonal"] = (g.norm(2, dim=-1, keepdim=True) - 1) ** 2
    fd["normal"] = F.normalize(g, dim=-1) * torch.tensor([1.0, -1.0, -1.0])
    fd["xyz"] = xyz
    fd["xyz_cam"] = xyz_cam
    fd["depth"] = depth / P["logscale"].exp()
    if has_bones:
        fd["gauss_density"] = gauss_density(P, xyz, fr["rest_articulation"])
    return fd, deltas, {"valid": valid, "inds": inds}


def query_field_eval_bg(P, fr, hxy, n_depth=64, alpha=None):
    """Eval-mode NeRF.query_field of the background field (nerf.py:580-684; rigid backward warp, get_valid_idx returns None
    for category "bg" (nerf.py:524-526), train-only fields return {}): importance sampling, rgb / density, visibility,
    normals through the rigid transform."""
    codes = {"basefield": fr["code_base"], "colorfield": fr["code_color"]}
    nc = n_depth // 2
    xyz_cam, _, deltas, depth = sample_cam_rays(hxy, fr["Kinv"], fr["near_far"], n_depth=nc)
    xyz, _ = cam_to_field(xyz_cam, None, fr["field2cam"])
    density = nerf_forward(P, xyz, codes, with_color=False, alpha=alpha, cfg=BG_CFG)
    weights, _ = compute_weights(density, deltas)
    depth_mid = (0.5 * (depth[:, :, :-1] + depth[:, :, 1:])).view(-1, nc - 1)
    new, inds = sample_pdf(depth_mid, weights.view(-1, nc)[:, 1:-1], nc, return_inds=True)
    depth_all, _ = torch.sort(torch.cat([depth, new.reshape(depth.shape)], -2), -2)
    xyz_cam, dir_cam, deltas, depth = sample_cam_rays(hxy, fr["Kinv"], fr["near_far"], depth=depth_all)
    xyz, dirs = cam_to_field(xyz_cam, dir_cam, fr["field2cam"])
    vis = vis_field(P, xyz, fr["code_vis"])
    rgb, density = nerf_forward(P, xyz, codes, alpha=alpha, cfg=BG_CFG, dir=dirs)
    fd = {"rgb": rgb, "density": density, "density_bg": density, "vis": vis}
    # NeRF.cycle_loss (nerf.py:905-925) is not train-only: a rigid field reports zeros
    for k in ("cyc_dist", "delta_skin", "skin_entropy"):
        fd[k] = torch.zeros_like(density)

    def fn_sdf(xc):
        xx, _ = cam_to_field(xc, None, fr["field2cam"])
        return nerf_forward(P, xx, codes, with_color=False, get_density=False, alpha=alpha, cfg=BG_CFG)

    with torch.enable_grad():
        xc = xyz_cam.detach().requires_grad_(True)
        sv = fn_sdf(xc)
        (g,) = torch.autograd.grad(sv, xc, torch.ones_like(sv))
    fd["eikonal"] = (g.norm(2, dim=-1, keepdim=True) - 1) ** 2
    fd["normal"] = F.normalize(g, dim=-1) * torch.tensor([1.0, -1.0, -1.0])
    fd["xyz"] = xyz
    fd["xyz_cam"] = xyz_cam
    fd["depth"] = depth / P["logscale"].exp()
    return fd, deltas, {"inds": inds}


def compute_eikonal_bg(P, xyz, code, rand_inds, alpha=None):
    """NeRF.compute_eikonal (nerf.py:416-453) for the bg field on the host-drawn ray subset."""
    M, N, D, _ = xyz.shape
    pts = xyz.reshape(M * N, D, 3)
    c = code[:, None, :].expand(M, N, code.shape[-1]).reshape(M * N, -1)
    out = torch.zeros(M * N, D, dtype=xyz.dtype)
    if rand_inds is None:
        rand_inds = torch.arange(M * N)

    def fn(x):
        return nerf_forward(P, x, {"basefield": c[rand_inds]}, with_color=False, get_density=False, alpha=alpha, cfg=BG_CFG)

    with torch.enable_grad():
        x = pts[rand_inds].detach().requires_grad_(True)
        sv = fn(x)
        (g,) = torch.autograd.grad(sv, x, torch.ones_like(sv), create_graph=True)
    out[rand_inds] = (g.norm(2, dim=-1) - 1) ** 2
    return out.reshape(M, N, D, 1)


def query_field_train_bg(P, fr, hxy, rng, flow_thresh=None, n_depth=64, alpha=None):
    """Training-mode NeRF.query_field of the background field (nerf.py:580-684): rigid warps, flow into the pair partner's
    camera (nerf.py:948-997), zero cycle terms, eikonal on the host-drawn ray subset."""
    codes = {"basefield": fr["code_base"], "colorfield": fr["code_color"]}
    xyz_cam, dir_cam, deltas, depth = sample_cam_rays(hxy, fr["Kinv"], fr["near_far"], n_depth=n_depth)
    xyz, dirs = cam_to_field(xyz_cam, dir_cam, fr["field2cam"])
    fd = {}
    vis = vis_field(P, xyz, fr["code_vis"])
    rgb, density = nerf_forward(P, xyz, codes, alpha=alpha, cfg=BG_CFG, dir=dirs)
    fd["rgb"], fd["density"], fd["density_bg"], fd["vis"] = rgb, density, density, vis
    nxt = flip_pair({k: fr[k] for k in ["Kinv", "field2cam"]})
    xyz_cam_next = field_to_cam(xyz, nxt["field2cam"])
    hxy_next = pinhole_projection(kmatinv(nxt["Kinv"]), xyz_cam_next)
    flow = (hxy_next - hxy.unsqueeze(-2))[..., :2]
    valid = xyz_cam_next[..., -1:] > 1e-6
    if flow_thresh is not None:
        valid = valid & (flow.norm(dim=-1, keepdim=True) < float(flow_thresh))
    fd["flow"] = torch.cat([flow, valid.to(flow.dtype)], -1)
    for k in ("cyc_dist", "delta_skin", "skin_entropy"):
        fd[k] = torch.zeros_like(density)
    fd["eikonal"] = compute_eikonal_bg(P, xyz, fr["code_base"], rng.get("eik_inds_bg"), alpha)
    fd["xyz"] = xyz
    fd["xyz_cam"] = xyz_cam
    fd["depth"] = depth / P["logscale"].exp()
    return fd, deltas, {}


def render_train_comp(P_fg, fr_fg, P_bg, fr_bg, hxy, rng, flow_thresh=None, n_depth=64):
    """dvr_model.render_samples for field_type == "comp" in training mode (engine/model.py:328-361)."""
    fd_fg, d_fg, aux = query_field_train(P_fg, fr_fg, hxy, rng, flow_thresh, n_depth)
    fd_bg, d_bg, _ = query_field_train_bg(P_bg, fr_bg, hxy, rng, flow_thresh, n_depth)
    fd, deltas = compose_fields({"fg": fd_fg, "bg": fd_bg}, {"fg": d_fg, "bg": d_bg})
    rendered = render_pixel(fd, deltas)
    aux_fg = dict(aux)
    aux_fg.update(render_pixel(fd_fg, d_fg))
    rendered["xyz_matches"] = aux["xyz_matches"]
    rendered["xyz_reproj"] = aux["xyz_reproj"]
    return {"rendered": rendered, "aux_dict": {"fg": aux_fg, "bg": render_pixel(fd_bg, d_bg)}}


def render_eval_comp(P_fg, fr_fg, P_bg, fr_bg, hxy, n_depth=64):
    """dvr_model.render_samples for field_type == "comp" in eval mode (engine/model.py:328-361): query both fields,
    compose_fields, render_pixel of the composite and of each field."""
    fd_fg, d_fg, dbg_fg = query_field_eval(P_fg, fr_fg, hxy, n_depth)
    fd_bg, d_bg, dbg_bg = query_field_eval_bg(P_bg, fr_bg, hxy, n_depth)
    fd, deltas = compose_fields({"fg": fd_fg, "bg": fd_bg}, {"fg": d_fg, "bg": d_bg})
    return {"rendered": render_pixel(fd, deltas), "aux_dict": {"fg": render_pixel(fd_fg, d_fg), "bg": render_pixel(fd_bg, d_bg)},
            "composed": fd, "deltas": deltas, "debug": {"fg": dbg_fg, "bg": dbg_bg}}


def render_eval(P, fr, hxy, n_depth=64, alpha=None):
    fd, deltas, aux = query_field_eval(P, fr, hxy, n_depth, alpha)
    out = render_pixel(fd, deltas)
    return {"rendered": out, "aux_dict": {"fg": dict(out)}, "debug": aux}


# ----------------------------------------------------------------------------
# compose_fields (fg+bg)                        (nnutils/multifields.py:339-398)
# ----------------------------------------------------------------------------


def compose_fields(multifields_dict, deltas_dict):
    """MultiFields.compose_fields: concatenate along D, zero-fill missing keys, sort by depth
    (stable argsort order must be reproduced), gather every key."""
    if len(multifields_dict) == 1:
        k = list(multifields_dict.keys())[0]
        return multifields_dict[k], deltas_dict[k]
    all_keys = []
    for fd in multifields_dict.values():
        for k in fd:
            if k not in all_keys:
                all_keys.append(k)
    cat = {}
    for k in all_keys:
        parts = []
        for cate, fd in multifields_dict.items():
            if k in fd:
                parts.append(fd[k])
            else:
                ref = [o[k] for o in multifields_dict.values() if k in o][0]
                parts.append(torch.zeros_like(ref))  # multifields.py:383-389
        cat[k] = torch.cat(parts, 2)
    deltas = torch.cat([deltas_dict[c] for c in multifields_dict], 2)
    # z-sort (multifields.py:387).  stable=True fixes the order of exactly equal depths (concatenation order), which the
    # reference leaves to torch.argsort's implementation; for distinct depths it is the same permutation
    order = cat["depth"].argsort(dim=2, stable=True)
    out = {k: torch.gather(v, 2, order.expand(-1, -1, -1, v.shape[-1])) for k, v in cat.items()}
    deltas = torch.gather(deltas, 2, order)
    return out, deltas


# ----------------------------------------------------------------------------
# losses                                           (engine/model.py:401-611)
# ----------------------------------------------------------------------------


def mask_balance_wt(mask, vis2d, is_detected):
    """dvr_model.get_mask_balance_wt (model.py:401-424)."""
    mask = mask.float()
    vis2d = vis2d.float() * is_detected.float()[:, None, None]
    if mask.sum() > 0 and (1 - mask).sum() > 0:
        pos = vis2d.sum() / mask[vis2d > 0].sum()
        neg = vis2d.sum() / (1 - mask[vis2d > 0]).sum()
        return 0.5 * pos * mask + 0.5 * neg * (1 - mask)
    return 1


def recon_losses_fg(results, batch, train_res, weights=None):
    """compute_recon_loss + mask_losses + apply_loss_weights for field_type == "fg"
    (model.py:426-501, 528-611), plus the rendered regularisers compute_reg_loss reads from
    `rendered`/`aux_dict` (model.py:503-526).  Returns dict of scalar losses."""
    r, a = results["rendered"], results["aux_dict"]["fg"]
    L = {}
    L["mask"] = (r["mask"] - batch["mask"].float()).pow(2) * mask_balance_wt(batch["mask"], batch["vis2d"], batch["is_detected"])
    L["feature"] = (a["feature"] - batch["feature"]).norm(2, -1, keepdim=True)
    L["feat_reproj"] = (a["xy_reproj"] - batch["hxy"][..., :2]).norm(2, -1, keepdim=True)
    L["rgb"] = (r["rgb"] - batch["rgb"]).pow(2)
    L["depth"] = (r["depth"] - batch["depth"]).norm(2, -1, keepdim=True)
    L["flow"] = (r["flow"] - batch["flow"]).norm(2, -1, keepdim=True) * (batch["flow_uct"] > 0).float()
    L["vis"] = a["vis"]
    if "gauss_mask" in r:  # model.py:497-501: SkinningWarp fields only
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
        v = v[v > 0].mean()
        if k in ("flow", "feat_reproj"):
            v = v / train_res
        if weights is not None and k + "_wt" in weights:
            v = v * weights[k + "_wt"]
        out[k] = v
    return out


def recon_losses_comp(results, batch, train_res, weights=None):
    """compute_recon_loss + mask_losses + apply_loss_weights for field_type == "comp" (model.py:426-501, 528-611): the fg mask
    is the rendered mask_fg, the composite must be opaque ((mask - 1)^2), visibility is supervised per field (bg at 1 %),
    dense terms are masked by vis2d only."""
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
        v = v[v > 0].mean()
        if k in ("flow", "feat_reproj"):
            v = v / train_res
        if weights is not None and k + "_wt" in weights:
            v = v * weights[k + "_wt"]
        out[k] = v
    return out


# loss weights: lab4d/config.py:20-46 defaults
DEFAULT_LOSS_WT = {
    "mask_wt": 0.1, "rgb_wt": 0.1, "depth_wt": 1e-4, "flow_wt": 0.5, "vis_wt": 1e-2,
    "feature_wt": 1e-2, "feat_reproj_wt": 5e-2, "reg_visibility_wt": 1e-4, "reg_eikonal_wt": 1e-3,
    "reg_deform_cyc_wt": 0.01, "reg_delta_skin_wt": 5e-3, "reg_skin_entropy_wt": 5e-4,
    "reg_gauss_skin_wt": 1e-3, "reg_cam_prior_wt": 0.1, "reg_skel_prior_wt": 0.1,
    "reg_gauss_mask_wt": 0.01, "reg_soft_deform_wt": 100.0,
}


# ----------------------------------------------------------------------------
# proxy-geometry refresh: nnutils/nerf.py:303-376, utils/geom_utils.py:344-362,392-476
# ----------------------------------------------------------------------------
def sample_grid(aabb, grid_size):
    """geom_utils.sample_grid (geom_utils.py:392-406)."""
    ax = [torch.linspace(float(aabb[0][i]), float(aabb[1][i]), grid_size) for i in range(3)]
    return torch.cartesian_prod(*ax)


def grid_query(P, aabb, grid_size, code_base, code_vis, extend=0.5):
    """The volume extract_canonical_mesh hands to marching cubes (nerf.py:327-343, geom_utils.py:467-476): sdf and visibility > 0
    on the dense grid of extend_aabb(aabb, extend), one instance code for every point."""
    box = extend_aabb(aabb, extend) if extend else aabb
    pts = sample_grid(box, grid_size)
    n = pts.shape[0]
    sdf = nerf_forward(P, pts[None], {"basefield": code_base}, with_color=False, get_density=False)[0]
    vis = vis_field(P, pts[None], code_vis)[0] > 0
    G = grid_size
    return sdf.view(G, G, G), vis.view(G, G, G), box


def get_near_far(pts, quat, trans, tol_fac=1.5):
    """geom_utils.get_near_far (geom_utils.py:344-362), cameras as (quat, trans)."""
    z = quaternion_translation_apply(quat[:, None].expand(-1, pts.shape[0], -1), trans[:, None].expand(-1, pts.shape[0], -1),
                                     pts[None].expand(quat.shape[0], -1, -1))[..., 2]
    pmax, pmin = z.max(-1)[0], z.min(-1)[0]
    delta = (pmax - pmin) * (tol_fac - 1)
    return torch.stack([pmin - delta, pmax + delta], -1).clamp(min=1e-3)
