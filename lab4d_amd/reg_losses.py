"""Host-side mirror of the regularisation terms of dvr_model.compute_reg_loss (lab4d/engine/model.py:503-526; SURVEY.md 8f
row 2): small extra queries of the same device fields on random points, plus the two priors on the per-frame modules.

  reg_visibility  NeRF.visibility_decay_loss            nnutils/nerf.py:396-414
  reg_gauss_skin  Deformable.gauss_skin_consistency_loss nnutils/deformable.py:200-236
  reg_soft_deform Deformable.soft_deform_loss            nnutils/deformable.py:238-252, warping.py:485-503
  reg_skel_prior  ArticulationSkelMLP.skel_prior_loss    nnutils/pose.py:575-590
  reg_cam_prior   CameraMLP.compute_distance_to_prior    nnutils/time.py:96-105, pose.py:84-90
(reg_eikonal / reg_deform_cyc / reg_delta_skin / reg_skin_entropy are rendered quantities of the main graph.)

The reference draws its random points and ids inside each term; here they are arguments (uniform draws `u` in [0,1)^3, ids),
so that a caller -- and the parity tests -- control the randomness, like `rng` in deformable.render_train.  Every field
evaluation runs on the gfx950 chain kernels (VisField, basefield, DenseWarp) / the gaussian-bone kernel; a point set is
queried as "one frame of n samples" when it shares one conditioning code and as "n frames of one sample" otherwise.
"""
import torch
import torch.nn.functional as F

from . import deformable as DF
from . import mlp, pose
from . import quat_utils as Q
from .warping import dense_warp


def sample_points_aabb(aabb, u, extend_factor=1.0):
    """NeRF.sample_points_aabb (nerf.py:378-394): u (n,3) uniform in [0,1) -> points in the extended aabb."""
    box = DF.extend_aabb(aabb, extend_factor)
    return u * (box[1:] - box[:1]) + box[:1]


def _as_frames(pts, code):
    """(n,3) points -> (M, spf, 3) with M = rows of the conditioning code (1: shared, n: one per point)."""
    if code.shape[0] == 1:
        return pts[None]
    if code.shape[0] != pts.shape[0]:
        raise RuntimeError("reg loss: %d codes for %d points" % (code.shape[0], pts.shape[0]))
    return pts[:, None]


def visibility_decay_loss(P, pts, code_vis, prec=mlp.PREC_F32):
    """-logsigmoid(-vis) averaged over the points.  code_vis: (1,32) or one instance code per point (n,32)."""
    vis = DF.vis_field(P, _as_frames(pts, code_vis), {"code_vis": code_vis}, prec)
    return -F.logsigmoid(-vis).mean()


def gauss_skin_consistency_loss(P, pts, rest_articulation_mean, code_base_mean, prec=mlp.PREC_F32):
    """Weighted BCE between the gaussian-bone occupancy at `pts` and the (detached) field occupancy density / ibeta.
    rest_articulation_mean ((1,B,4),(1,B,4)) = articulation.get_mean_vals(); code_base_mean (1,32) = mean instance code."""
    _, centre = Q.dual_quaternion_to_quaternion_translation(rest_articulation_mean)
    one = torch.ones(1, device=pts.device)
    density_gauss = DF.GaussDensity.apply(pts.contiguous(), centre[0], one)  # warping.py:355-387: no ibeta factor here
    with torch.no_grad():
        density = DF.nerf_forward(P, pts[None], {"code_base": code_base_mean}, prec, with_color=False)[0] / P["logibeta"].exp()
        weight_pos = 0.5 / (1e-6 + density.mean())
        weight_neg = 0.5 / (1e-6 + 1 - density).mean()
        weight = density * weight_pos + (1 - density) * weight_neg
    return F.binary_cross_entropy(density_gauss, density, weight=weight)


def soft_deform_loss(P, pts, t_embed, code_fw, code_bw, prec=mlp.PREC_F32):
    """ComposedWarp.compute_post_warp_dist2 averaged: |dense_fw(x) - x|^2 and the forward/backward cycle of the post-warp.
    t_embed (n,128): post-warp time embedding of each point's frame; code_fw / code_bw (n,32): instance codes."""
    x = pts[:, None, None]
    x_t = dense_warp(P, x, t_embed, code_fw, False, prec)
    dist2 = (x_t - x).pow(2).sum(-1)
    x_back = dense_warp(P, x_t, t_embed, code_bw, True, prec)
    return ((dist2 + (x_t - x_back).pow(2).sum(-1)) * 0.5).mean()


def skel_prior_loss(P, prefix, info):
    """Rest-pose joint angles and mean log-bone-length increments pulled to zero."""
    so3 = pose.articulation_so3(P, prefix, pose.time_embedding_mean(P, prefix + ".time_embedding", info))
    inc = pose.log_bone_len(P, prefix + ".log_bone_len", None, 1)
    return so3.pow(2).mean() + 0.02 * inc.pow(2).mean()


def quaternion_translation_to_se3(q, t):
    """quat_transform.py:217-252,305-310: (M,4),(M,3) -> (M,4,4)."""
    w, x, y, z = q.unbind(-1)
    s = 2.0 / (q * q).sum(-1)
    R = torch.stack([1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w),
                     s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w),
                     s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)], -1).reshape(q.shape[:-1] + (3, 3))
    bot = torch.tensor([0.0, 0, 0, 1], device=q.device).expand(q.shape[:-1] + (1, 4))
    return torch.cat([torch.cat([R, t[..., None]], -1), bot], -2)


def cam_prior_loss(P, prefix, info, init_vals):
    """mse between the camera SE(3) of all frames and the initial cameras (N,4,4)."""
    q, t = pose.camera_vals(P, prefix, None, info)
    return F.mse_loss(quaternion_translation_to_se3(q, t), init_vals)
