"""CPU oracle for the per-frame pose / articulation path.

TEST INFRASTRUCTURE ONLY -- never imported by the product path (lab4d_amd/); see the header of lab4d_oracle.py.

torch-CPU fp32 restatement of `TimeEmbedding` -> `TimeMLP` -> `CameraMLP` / `ArticulationSkelMLP` and of the
forward-kinematics helpers of lab4d/utils/skel_utils.py; each function cites the reference file:line it follows.
Weights live in a flat dict keyed by the reference's state_dict names under a prefix.  The skeleton tables
(`rest_joints`, `edges`, `symm_idx`, skel_utils.py:183-357) are *data*: they come from the reference module / the golden
fixture, not from this file.

Parity pinning: tests/test_oracle_golden.py::test_pose_* against tests/golden/pose.pt, which
tests/golden/make_golden.py generates from the reference's own modules.
"""
import torch
import torch.nn.functional as F

from .lab4d_oracle import base_mlp, pos_embedding, quaternion_mul


# ----------------------------------------------------------------------------
# rotation conversions                 (utils/geom_utils.py, utils/quat_transform.py)
# ----------------------------------------------------------------------------


def hat_map(v):
    """geom_utils.py:86-107."""
    z = torch.zeros_like(v[..., 0])
    rows = [torch.stack([z, -v[..., 2], v[..., 1]], -1), torch.stack([v[..., 2], z, -v[..., 0]], -1),
            torch.stack([-v[..., 1], v[..., 0], z], -1)]
    return torch.stack(rows, -2)


def so3_to_exp_map(so3, eps=1e-6):
    """geom_utils.py:110-140: Rodrigues with theta clamped from below (not a Taylor branch)."""
    theta = torch.clamp(so3.norm(dim=-1, keepdim=True), eps)
    V = hat_map(so3 / theta)
    theta = theta[..., None]
    return torch.eye(3, dtype=so3.dtype) + torch.sin(theta) * V + (1 - torch.cos(theta)) * (V @ V)


def axis_angle_to_quaternion(aa):
    """quat_transform.py:149-174."""
    ang = aa.norm(dim=-1, keepdim=True)
    k = torch.where(ang.abs() < 1e-6, 0.5 - ang * ang / 48, torch.sin(ang * 0.5) / ang)
    return torch.cat([torch.cos(ang * 0.5), aa * k], -1)


def matrix_to_quaternion(m):
    """quat_transform.py:468-532: four candidates (the quaternion times each of r,i,j,k), pick the best conditioned one
    (largest |component|, first on ties); the picked component comes out positive."""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.reshape(m.shape[:-2] + (9,)).unbind(-1)
    s = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1)
    pos = s > 0
    q_abs = torch.where(pos, torch.sqrt(torch.where(pos, s, torch.ones_like(s))), torch.zeros_like(s))  # zero subgradient at 0
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], -1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], -1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], -1)], -2)
    cand = cand / (2.0 * q_abs[..., None].clamp(min=0.1))
    pick = q_abs.argmax(-1)
    return torch.gather(cand, -2, pick[..., None, None].expand(pick.shape + (1, 4)))[..., 0, :]


def quaternion_translation_to_dual_quaternion(q, t):
    """quat_transform.py:290-297."""
    return q, 0.5 * quaternion_mul(t, q)


# ----------------------------------------------------------------------------
# skeleton forward kinematics                     (utils/skel_utils.py)
# ----------------------------------------------------------------------------


def valid_edges(edges):
    """skel_utils.py:18-32: (child, parent) 0-based pairs of the edges whose parent is a joint (not the root 0)."""
    pairs = [(c - 1, p - 1) for c, p in edges.items() if p > 0]
    return [c for c, _ in pairs], [p for _, p in pairs]


def rest_joints_to_local(rest_joints, edges):
    """skel_utils.py:35-47."""
    idx, parent = valid_edges(edges)
    local = rest_joints.clone()
    local[idx] = rest_joints[idx] - rest_joints[parent]
    return local


def fk_global(local_rest_joints, so3, edges):
    """skel_utils.py:50-94 (to_dq=False part): global (R, t) of every joint.  The reference walks `edges` in dict order and
    reads the parent's *current* global transform (identity until the parent itself has been visited)."""
    R_loc = so3_to_exp_map(so3)
    B = so3.shape[-2]
    eye = torch.eye(3, dtype=so3.dtype).expand(so3.shape[:-2] + (3, 3))
    zero = torch.zeros_like(so3[..., 0, :])
    G_R, G_t = [eye] * B, [zero] * B
    for idx, par in edges.items():
        j = idx - 1
        P_R, P_t = (G_R[par - 1], G_t[par - 1]) if par > 0 else (eye, zero)
        G_R[j] = P_R @ R_loc[..., j, :, :]
        G_t[j] = (P_R @ local_rest_joints[..., j, :, None])[..., 0] + P_t
    return torch.stack(G_R, -3), torch.stack(G_t, -2)


def fk_se3(local_rest_joints, so3, edges):
    """skel_utils.py:50-103 with to_dq=True."""
    G_R, G_t = fk_global(local_rest_joints, so3, edges)
    return quaternion_translation_to_dual_quaternion(matrix_to_quaternion(G_R), G_t)


def shift_joints_to_bones(joints, edges):
    """skel_utils.py:127-145: a joint with children moves to the mean of the midpoints to its children; leaves stay."""
    idx, parent = valid_edges(edges)
    mid = 0.5 * (joints[..., parent, :] + joints[..., idx, :])
    out = [joints[..., j, :] for j in range(joints.shape[-2])]
    for p in sorted(set(parent)):
        sel = [k for k, pp in enumerate(parent) if pp == p]
        out[p] = mid[..., sel[0], :] if len(sel) == 1 else mid[..., sel, :].mean(-2)
    return torch.stack(out, -2)


def shift_joints_to_bones_dq(dq, edges, shift=None):
    """skel_utils.py:106-124."""
    qr, qd = dq
    t = 2 * quaternion_mul(qd, qr * torch.tensor([1.0, -1, -1, -1]))[..., 1:]
    if shift is not None:
        t = t + shift
    return quaternion_translation_to_dual_quaternion(qr, shift_joints_to_bones(t, edges))


# ----------------------------------------------------------------------------
# per-frame modules                       (nnutils/embedding.py, time.py, pose.py)
# ----------------------------------------------------------------------------


def frame_tid(frame_id, info):
    """embedding.py:177-184: normalised time in [-1, 1] inside the frame's video."""
    fid = frame_id.long()
    sub = frame_id - info["raw_fid_to_vstart"][fid]
    return (sub - info["raw_fid_to_vidlen"][fid] / 2) / info["max_ts"] * 2 * info.get("time_scale", 1.0)


def time_embedding(P, prefix, frame_id, info):
    """TimeEmbedding.forward (embedding.py:194-217): Fourier(t) -> mapping1 ; cat video code ; mapping2."""
    if frame_id is None:
        inst_id, t = info["frame_to_vid"], frame_tid(info["frame_mapping"], info)
    else:
        inst_id, t = info["raw_fid_to_vid"][frame_id], frame_tid(frame_id, info)
    coeff = pos_embedding(t[..., None].float(), info["num_freq_t"])
    coeff = F.linear(coeff, P[f"{prefix}.mapping1.weight"], P[f"{prefix}.mapping1.bias"])
    w = P[f"{prefix}.inst_embedding.mapping.weight"]
    code = w[torch.zeros_like(inst_id) if w.shape[0] == 1 else inst_id]
    return F.linear(torch.cat([coeff, code], -1), P[f"{prefix}.mapping2.weight"], P[f"{prefix}.mapping2.bias"])


def time_embedding_mean(P, prefix, info):
    """TimeEmbedding.get_mean_embedding (embedding.py:219-227)."""
    return time_embedding(P, prefix, info["frame_mapping"], info).mean(0, keepdim=True)


def time_mlp(P, prefix, t_embed, D=5):
    """TimeMLP.forward (time.py:65-73): BaseMLP(D=5, W=256, skips=[], final_act=True)."""
    return base_mlp(P, prefix, t_embed, D, skips=(), final_act=True)


def _head(P, prefix, x):
    """nn.Sequential(Linear(W, W//2), ReLU, Linear(W//2, C)) (pose.py:69-78,281-285,371-375)."""
    h = F.relu(F.linear(x, P[f"{prefix}.0.weight"], P[f"{prefix}.0.bias"]))
    return F.linear(h, P[f"{prefix}.2.weight"], P[f"{prefix}.2.bias"])


def camera_vals(P, prefix, frame_id, info):
    """CameraMLP.get_vals (pose.py:116-147): (quat (M,4), trans (M,3)); quat = normalize(head) * normalize(base_quat[vid])."""
    feat = time_mlp(P, prefix, time_embedding(P, f"{prefix}.time_embedding", frame_id, info))
    trans = _head(P, f"{prefix}.trans", feat)
    quat = F.normalize(_head(P, f"{prefix}.quat", feat), dim=-1)
    inst_id = info["frame_to_vid"] if frame_id is None else info["raw_fid_to_vid"][frame_id]
    return quaternion_mul(quat, F.normalize(P[f"{prefix}.base_quat"][inst_id], dim=-1)), trans


def log_bone_len(P, prefix, inst_id, rows):
    """CondMLP(num_inst, in_channels=0, D=2, W=64) (pose.py:381-388, base.py:123-150): the input is the instance code alone;
    inst_id None -> mean code."""
    w = P[f"{prefix}.inst_embedding.mapping.weight"]
    if inst_id is None:
        code = w.mean(0).expand(rows, -1)
    else:
        code = w[torch.zeros_like(inst_id) if w.shape[0] == 1 else inst_id]
    return base_mlp(P, prefix, code, 2, skips=(4,))


def rel_rest_joints(P, prefix, skel, inst_id=None):
    """ArticulationSkelMLP.compute_rel_rest_joints (pose.py:472-502): local rest joints scaled by the symmetrised bone length."""
    local = rest_joints_to_local(skel["rest_joints"], skel["edges"])[None]
    rows = 1 if inst_id is None else inst_id.shape[0]
    local = local.expand(rows, -1, -1)
    length = (log_bone_len(P, f"{prefix}.log_bone_len", inst_id, rows) + P[f"{prefix}.logscale"]).exp()
    length = (length + length[..., skel["symm_idx"]]) / 2
    return local * length[..., None]


def articulation_so3(P, prefix, t_embed):
    """pose.py:442-447."""
    so3 = _head(P, f"{prefix}.so3", time_mlp(P, prefix, t_embed))
    return so3.reshape(t_embed.shape[:-1] + (-1, 3))


def articulation_skel_forward(P, prefix, skel, t_embed, inst_id, local_rest_joints=None):
    """ArticulationSkelMLP.forward (pose.py:417-470)."""
    so3 = articulation_so3(P, prefix, t_embed)
    if local_rest_joints is None:
        local_rest_joints = rel_rest_joints(P, prefix, skel, inst_id)
    dq = fk_se3(local_rest_joints.expand_as(so3), so3, skel["edges"])
    return shift_joints_to_bones_dq(dq, skel["edges"], shift=P[f"{prefix}.shift"])


def articulation_skel_vals_and_mean(P, prefix, skel, frame_id, info):
    """ArticulationSkelMLP.get_vals_and_mean (pose.py:526-573): one batched FK over [frames ; rest pose]."""
    inst_id = info["frame_to_vid"] if frame_id is None else info["raw_fid_to_vid"][frame_id]
    bs = inst_id.shape[0]
    te = time_embedding(P, f"{prefix}.time_embedding", frame_id, info)
    te_mean = time_embedding_mean(P, f"{prefix}.time_embedding", info).expand(bs, -1)
    rel = torch.cat([rel_rest_joints(P, prefix, skel, inst_id), rel_rest_joints(P, prefix, skel).expand(bs, -1, -1)], 0)
    qr, qd = articulation_skel_forward(P, prefix, skel, torch.cat([te, te_mean], 0), None, local_rest_joints=rel)
    return (qr[:bs], qd[:bs]), (qr[bs:], qd[bs:])


def articulation_flat_forward(P, prefix, t_embed):
    """ArticulationFlatMLP.forward (pose.py:287-303), bag-of-bones motion: per-bone translation (x 0.1, ScaleLayer) and axis-angle
    heads on the time feature -> dual quaternions."""
    feat = time_mlp(P, prefix, t_embed)
    trans = (_head(P, f"{prefix}.trans", feat) * 0.1).reshape(t_embed.shape[:-1] + (-1, 3))
    so3 = _head(P, f"{prefix}.so3", feat).reshape(t_embed.shape[:-1] + (-1, 3))
    return quaternion_translation_to_dual_quaternion(axis_angle_to_quaternion(so3), trans)


def intrinsics_vals(P, prefix, frame_id, info):
    """IntrinsicsMLP.get_vals (intrinsics.py:86-107): (M,4) = [fx, fy, px, py]; focal = exp(head) * exp(base_logfocal[vid]),
    averaged over x/y (square pixels); principal point = base_ppoint[vid].  info: the module's own TimeEmbedding tables
    (num_freq_t = 0, time_scale = 0.1 by default)."""
    focal = _head(P, f"{prefix}.focal", time_mlp(P, prefix, time_embedding(P, f"{prefix}.time_embedding", frame_id, info))).exp()
    inst_id = info["frame_to_vid"] if frame_id is None else info["raw_fid_to_vid"][frame_id]
    focal = focal * P[f"{prefix}.base_logfocal"][inst_id].exp()
    focal = (focal + focal.flip(-1)) / 2
    return torch.cat([focal, P[f"{prefix}.base_ppoint"][inst_id].expand_as(focal)], -1)
