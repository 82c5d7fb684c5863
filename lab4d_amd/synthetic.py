"""Seeded synthetic weights, per-frame inputs and rays for tests and bench.py.

There is no dataset or checkpoint offline, so everything is generated:
  * weights: the per-sample parameters of the reference's fg field
    `Deformable("skel-quad", num_freq_dir=-1, appr_channels=32, num_inst=1, init_scale=0.2)`
    (lab4d/nnutils/multifields.py:77-84) with the reference's own state_dict names and
    shapes, default-PyTorch-style init (U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for Linear,
    N(0,1) for Embedding) drawn from a seeded CPU generator;
  * per-frame inputs: camera pose, intrinsics, near/far, bone articulations and the
    per-frame codes that the reference's per-frame MLPs (outside the hot path) would produce.
Pure torch-CPU; results are moved to the device by the caller.
"""
import math

import torch

NUM_BONES = 25  # skel-quad (lab4d/utils/skel_utils.py quad skeleton)
# left<->right bone permutation of the quad skeleton (skel_utils.py:349-357 QUAD_SYMM_IDX, 0-based);
# SkinningField.get_gauss averages log_gauss over it (skinning.py:142-153)
QUAD_SYMM_IDX = [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15, 16, 21, 22, 23, 24, 17, 18, 19, 20]
# the 18-joint human skeleton (skel_utils.py:348-349 HUMAN_SYMM_IDX, 0-based): fg_motion "skel-human" (BASELINE configs[2])
HUMAN_SYMM_IDX = [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 15, 16, 17, 12, 13, 14]
SYMM_IDX = {25: QUAD_SYMM_IDX, 18: HUMAN_SYMM_IDX}

# (name, out, in) for every per-sample Linear on the fg hot path (SURVEY.md 8a notes)
FG_LINEARS = [
    ("basefield.linear_1.0", 256, 95), ("basefield.linear_2.0", 256, 256), ("basefield.linear_3.0", 256, 256),
    ("basefield.linear_4.0", 256, 256), ("basefield.linear_5.0", 256, 351), ("basefield.linear_6.0", 256, 256),
    ("basefield.linear_7.0", 256, 256), ("basefield.linear_8.0", 256, 256), ("basefield.linear_final.0", 256, 256),
    ("colorfield.linear_1.0", 256, 107), ("colorfield.linear_2.0", 256, 256), ("colorfield.linear_final.0", 256, 256),
    ("sdf", 1, 256), ("rgb.0", 128, 288), ("rgb.2", 3, 128),
    ("vis_mlp.basefield.linear_1.0", 64, 95), ("vis_mlp.basefield.linear_2.0", 64, 64),
    ("vis_mlp.basefield.linear_final", 1, 64),
    ("feature_field.linear_1.0", 128, 39), ("feature_field.linear_2.0", 128, 128),
    ("feature_field.linear_3.0", 128, 128), ("feature_field.linear_4.0", 128, 128),
    ("feature_field.linear_5.0", 128, 167), ("feature_field.linear_final", 16, 128),
    ("warp.skinning_model.delta_field.linear_1.0", 64, 235), ("warp.skinning_model.delta_field.linear_2.0", 64, 64),
    ("warp.skinning_model.delta_field.linear_final", NUM_BONES, 64),
]
FG_EMBEDDINGS = [
    ("basefield.inst_embedding.mapping.weight", 32), ("colorfield.inst_embedding.mapping.weight", 32),
    ("vis_mlp.basefield.inst_embedding.mapping.weight", 32),
    ("warp.skinning_model.delta_field.inst_embedding.mapping.weight", 32),
]


# ComposedWarp's dense post-warp (warping.py:445-447: DenseWarp(D=2, W=256)): two CondMLPs, 199 = 39 + 128 + 32 inputs
DENSE_LINEARS = [(f"warp.post_warp.{m}.linear_1.0", 256, 199) for m in ("forward_map", "backward_map")] + \
                [(f"warp.post_warp.{m}.linear_2.0", 256, 256) for m in ("forward_map", "backward_map")] + \
                [(f"warp.post_warp.{m}.linear_final", 3, 256) for m in ("forward_map", "backward_map")]
DENSE_EMBEDDINGS = [(f"warp.post_warp.{m}.inst_embedding.mapping.weight", 32) for m in ("forward_map", "backward_map")]


def add_dense_weights(P, seed=0, num_inst=1):
    """Adds the post-warp parameters of fg_motion "comp_skel-quad_dense" to a make_weights() dict (own generator, so the
    skel-quad weights and their golden checksums are unchanged)."""
    g = torch.Generator().manual_seed(seed + 7919)
    for name, o, i in DENSE_LINEARS:
        bound = 1.0 / math.sqrt(i)
        P[name + ".weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * bound
        P[name + ".bias"] = (torch.rand(o, generator=g) * 2 - 1) * bound
    for name, c in DENSE_EMBEDDINGS:
        P[name] = torch.randn(num_inst, c, generator=g)
    return P


# the background field NeRF(num_freq_xyz=6, num_freq_dir=0, appr_channels=0, D=5, W=128) (multifields.py:86-93); SURVEY 8a notes
BG_LINEARS = [
    ("basefield.linear_1.0", 128, 71), ("basefield.linear_2.0", 128, 128), ("basefield.linear_3.0", 128, 128),
    ("basefield.linear_4.0", 128, 128), ("basefield.linear_5.0", 128, 199), ("basefield.linear_final.0", 128, 128),
    ("colorfield.linear_1.0", 128, 83), ("colorfield.linear_2.0", 128, 128), ("colorfield.linear_final.0", 128, 128),
    ("sdf", 1, 128), ("rgb.0", 64, 131), ("rgb.2", 3, 64),
    ("vis_mlp.basefield.linear_1.0", 64, 95), ("vis_mlp.basefield.linear_2.0", 64, 64), ("vis_mlp.basefield.linear_final", 1, 64),
]
BG_EMBEDDINGS = [("basefield.inst_embedding.mapping.weight", 32), ("colorfield.inst_embedding.mapping.weight", 32),
                 ("vis_mlp.basefield.inst_embedding.mapping.weight", 32)]


def make_bg_weights(seed=0, num_inst=1):
    """Per-sample parameters of the bg NeRF, keyed by its own state_dict names (no prefix)."""
    g = torch.Generator().manual_seed(seed + 15485863)
    P = {}
    for name, o, i in BG_LINEARS:
        bound = 1.0 / math.sqrt(i)
        P[name + ".weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * bound
        P[name + ".bias"] = (torch.rand(o, generator=g) * 2 - 1) * bound
    for name, c in BG_EMBEDDINGS:
        P[name] = torch.randn(num_inst, c, generator=g)
    P["logibeta"] = torch.tensor([-math.log(0.1)])
    P["logscale"] = torch.tensor([math.log(0.1)])
    return P


def make_bg_frames(seed, M, res):
    """Per-frame inputs of the background field: its own camera pose (scene-to-camera) and near/far."""
    g = torch.Generator().manual_seed(seed + 32452843)
    fr = {}
    K = torch.tensor([[res, 0, res / 2], [0, res, res / 2], [0, 0, 1]], dtype=torch.float32)
    fr["Kinv"] = torch.linalg.inv(K)[None].repeat(M, 1, 1).contiguous()
    q = _rand_unit_quat(g, M, 0.3)
    t = torch.tensor([0.0, 0.0, 0.9]).repeat(M, 1) + 0.02 * torch.randn(M, 3, generator=g)
    fr["field2cam"] = (q.contiguous(), t.contiguous())
    fr["near_far"] = torch.stack([t[:, 2] - 0.6, t[:, 2] + 0.6], -1).contiguous()
    fr["frame_id"] = torch.arange(M, dtype=torch.long)
    fr["inst_id"] = torch.zeros(M, dtype=torch.long)
    return fr


def add_bg_codes(fr, P):
    look = lambda name: P[name][torch.zeros_like(fr["inst_id"]) if P[name].shape[0] == 1 else fr["inst_id"]]
    fr["code_base"] = look("basefield.inst_embedding.mapping.weight")
    fr["code_color"] = look("colorfield.inst_embedding.mapping.weight")
    fr["code_vis"] = look("vis_mlp.basefield.inst_embedding.mapping.weight")
    return fr


# fg_motion "dense" (warping.py:37-38,94-141): DenseWarp with its class defaults D=6, W=256, skips=[4]; 199 = 39 + 128 + 32 inputs
DENSE6_LINEARS = [(f"warp.{m}.{l}", o, i) for m in ("forward_map", "backward_map")
                  for l, o, i in (("linear_1.0", 256, 199), ("linear_2.0", 256, 256), ("linear_3.0", 256, 256), ("linear_4.0", 256, 256),
                                  ("linear_5.0", 256, 455), ("linear_6.0", 256, 256), ("linear_final", 3, 256))]
DENSE6_EMBEDDINGS = [(f"warp.{m}.inst_embedding.mapping.weight", 32) for m in ("forward_map", "backward_map")]


def make_weights(seed=0, num_inst=1, sdf_bias=None, num_bones=NUM_BONES, motion="skinning"):
    """Flat dict of fp32 CPU tensors keyed by the reference's state_dict names.  motion: "skinning" (bob / skel-*), "rigid" (fg_motion
    "rigid", the reference's default: no warp parameters) or "dense" (fg_motion "dense": the two 6-layer maps of a bare DenseWarp); the
    non-warp parameters are the same numbers for every motion."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name, o, i in FG_LINEARS:
        if name.startswith("warp.skinning_model.delta_field") and num_bones != NUM_BONES:  # skinning.py:70-86: sized by the skeleton
            i = 3 * num_bones + 160 if name.endswith("linear_1.0") else i
            o = num_bones if name.endswith("linear_final") else o
        bound = 1.0 / math.sqrt(i)
        P[name + ".weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * bound
        P[name + ".bias"] = (torch.rand(o, generator=g) * 2 - 1) * bound
    for name, c in FG_EMBEDDINGS:
        P[name] = torch.randn(num_inst, c, generator=g)
    P["logibeta"] = torch.tensor([-math.log(0.1)])  # nerf.py:144-145
    P["logscale"] = torch.tensor([math.log(0.2)])  # nerf.py:147-148, init_scale=0.2
    P["logsigma"] = torch.tensor([0.0])  # feature.py:86-87
    P["warp.logibeta"] = torch.tensor([-math.log(0.01)])  # warping.py:273-275
    P["warp.skinning_model.log_gauss"] = torch.full((num_bones, 3), math.log(0.03))  # skinning.py:62-66
    P["warp.skinning_model.log_gauss"] += 0.1 * torch.randn(num_bones, 3, generator=g)
    P["warp.skinning_model.symm_idx"] = torch.tensor(SYMM_IDX[num_bones], dtype=torch.long)
    P["aabb"] = torch.tensor([[-0.12, -0.12, -0.12], [0.12, 0.12, 0.12]])  # proxy sphere r=0.12
    if sdf_bias is not None:
        P["sdf.bias"] = torch.tensor([float(sdf_bias)])
    if motion in ("rigid", "dense"):
        P = {k: v for k, v in P.items() if not k.startswith("warp.")}
    if motion == "dense":
        gd = torch.Generator().manual_seed(seed + 611953)
        for name, o, i in DENSE6_LINEARS:
            bound = 1.0 / math.sqrt(i)
            P[name + ".weight"] = (torch.rand(o, i, generator=gd) * 2 - 1) * bound
            P[name + ".bias"] = (torch.rand(o, generator=gd) * 2 - 1) * bound
        for name, c in DENSE6_EMBEDDINGS:
            P[name] = torch.randn(num_inst, c, generator=gd)
    return P


def _rand_unit_quat(g, n, angle):
    axis = torch.randn(n, 3, generator=g)
    axis = axis / axis.norm(dim=-1, keepdim=True)
    ang = angle * (torch.rand(n, 1, generator=g) * 2 - 1)
    return torch.cat([torch.cos(ang / 2), axis * torch.sin(ang / 2)], -1)


def _qmul(a, b):
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


def _qt_to_dq(q, t):
    # quaternion_translation_to_dual_quaternion (quat_transform.py:290-297): q_d = 0.5 * t * q
    tq = torch.cat([torch.zeros_like(t[..., :1]), t], -1)
    return q, 0.5 * _qmul(tq, q)


def make_frames(seed, M, res, num_inst=1, num_bones=NUM_BONES):
    """Per-frame inputs for M frames (M even: consecutive frames form a pair, nerf.py:929-946)."""
    g = torch.Generator().manual_seed(seed)
    fr = {}
    # intrinsics: focal = res, principal point = res/2; Kinv = K^-1
    K = torch.tensor([[res, 0, res / 2], [0, res, res / 2], [0, 0, 1]], dtype=torch.float32)
    fr["Kinv"] = torch.linalg.inv(K)[None].repeat(M, 1, 1).contiguous()
    # camera: small rotation about a random axis, object 0.6 in front (3.0 * init_scale 0.2)
    q = _rand_unit_quat(g, M, 0.5)
    t = torch.tensor([0.0, 0.0, 0.6]).repeat(M, 1) + 0.02 * torch.randn(M, 3, generator=g)
    fr["field2cam"] = (q.contiguous(), t.contiguous())
    fr["near_far"] = torch.stack([t[:, 2] - 0.18, t[:, 2] + 0.18], -1).contiguous()
    # bones: rest centres inside the r=0.1 ball, identity rest rotation; time-t = small motion
    centres = 0.06 * torch.randn(num_bones, 3, generator=g)
    rest_q = torch.tensor([1.0, 0, 0, 0]).repeat(M, num_bones, 1)
    rest_t = centres[None].repeat(M, 1, 1)
    fr["rest_articulation"] = tuple(x.contiguous() for x in _qt_to_dq(rest_q, rest_t))
    dq_q = _rand_unit_quat(g, M * num_bones, 0.6).view(M, num_bones, 4)
    t_t = rest_t + 0.01 * torch.randn(M, num_bones, 3, generator=g)
    fr["t_articulation"] = tuple(x.contiguous() for x in _qt_to_dq(dq_q, t_t))
    fr["t_embed"] = 0.5 * torch.randn(M, 128, generator=g)
    fr["t_embed_mean"] = 0.5 * torch.randn(1, 128, generator=g)
    fr["t_embed_dense"] = 0.5 * torch.randn(M, 128, generator=torch.Generator().manual_seed(seed + 104729))  # post-warp TimeEmbedding
    fr["appr_code"] = 0.5 * torch.randn(M, 32, generator=g)
    fr["frame_id"] = torch.arange(M, dtype=torch.long)
    fr["inst_id"] = torch.zeros(M, dtype=torch.long)
    return fr


def add_codes(fr, P):
    """Instance codes = InstEmbedding lookups (embedding.py:246-264), num_inst == 1 -> row 0."""
    def look(name):
        w = P[name]
        idx = fr["inst_id"] if w.shape[0] > 1 else torch.zeros_like(fr["inst_id"])
        return w[idx]
    fr["code_base"] = look("basefield.inst_embedding.mapping.weight")
    fr["code_color"] = look("colorfield.inst_embedding.mapping.weight")
    fr["code_vis"] = look("vis_mlp.basefield.inst_embedding.mapping.weight")
    if "warp.skinning_model.delta_field.inst_embedding.mapping.weight" in P:
        fr["code_skin"] = look("warp.skinning_model.delta_field.inst_embedding.mapping.weight")
    elif "warp.forward_map.inst_embedding.mapping.weight" in P:  # fg_motion "dense"
        fr["motion"] = "dense"
        fr["code_dense_fw"], fr["code_dense_bw"] = look("warp.forward_map.inst_embedding.mapping.weight"), look("warp.backward_map.inst_embedding.mapping.weight")
    else:
        fr["motion"] = "rigid"
    if "warp.post_warp.forward_map.inst_embedding.mapping.weight" in P:
        fr["dense"] = {"t_embed": fr["t_embed_dense"], "code_fw": look("warp.post_warp.forward_map.inst_embedding.mapping.weight"),
                       "code_bw": look("warp.post_warp.backward_map.inst_embedding.mapping.weight")}
    return fr


def make_rays(res, M, rows=None):
    """Full pixel grid hxy (M, N, 3) = (x+0.5?, ...) -- the reference's create_xy_grid
    (trainer.py:493-506) uses integer pixel coordinates [0,res) with homogeneous 1."""
    ys, xs = torch.meshgrid(torch.arange(res, dtype=torch.float32), torch.arange(res, dtype=torch.float32), indexing="ij")
    if rows is not None:  # (first, last+1) of a contiguous band, or an explicit list / tensor of row indices
        sel = slice(rows[0], rows[1]) if isinstance(rows, tuple) else torch.as_tensor(rows, dtype=torch.long)
        ys, xs = ys[sel], xs[sel]
    hxy = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(xs.numel())], -1)
    return hxy[None].repeat(M, 1, 1).contiguous()


def make_targets(seed, M, N, res, hxy):
    """Loss targets: rgb~U(0,1), mask = disc of radius res/4, depth=1, flow=0, ..."""
    g = torch.Generator().manual_seed(seed)
    c = res / 2
    b = {}
    b["rgb"] = torch.rand(M, N, 3, generator=g)
    b["mask"] = ((hxy[..., :2] - c).norm(dim=-1, keepdim=True) < res / 4)
    b["depth"] = torch.ones(M, N, 1)
    b["flow"] = torch.zeros(M, N, 2)
    b["flow_uct"] = torch.ones(M, N, 1)
    b["vis2d"] = torch.ones(M, N, 1)
    b["is_detected"] = torch.ones(M)
    f = torch.randn(M, N, 16, generator=g)
    b["feature"] = f / f.norm(dim=-1, keepdim=True)
    b["hxy"] = hxy
    return b


def to_device(x, device):
    if torch.is_tensor(x):
        return x.to(device)
    if isinstance(x, tuple):
        return tuple(to_device(t, device) for t in x)
    if isinstance(x, dict):
        return {k: to_device(v, device) for k, v in x.items()}
    return x
