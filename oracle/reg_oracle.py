"""CPU oracle for the regularisation terms of dvr_model.compute_reg_loss (engine/model.py:503-526):
small extra field queries on random points plus two priors on the per-frame modules.

TEST INFRASTRUCTURE ONLY -- never imported by the product path (lab4d_amd/); see the header of lab4d_oracle.py.

The reference draws the random points / ids inside each loss (torch.rand / torch.randint); here the draws are inputs, so
that the reference, this restatement and the device path can be fed identical numbers.

Parity pinning: tests/test_oracle_golden.py::test_reg_* against tests/golden/reg.pt (tests/golden/make_reg_golden.py runs the
reference's own Deformable("comp_skel-quad_dense") methods with the random draws replayed).
"""
import torch
import torch.nn.functional as F

from . import lab4d_oracle as O
from . import pose_oracle as PO


def sample_points_aabb(aabb, u, extend_factor=1.0):
    """NeRF.sample_points_aabb (nerf.py:378-394) with the uniform draws u (n,3) supplied."""
    lo_hi = O.extend_aabb(aabb, extend_factor)
    return u * (lo_hi[1:] - lo_hi[:1]) + lo_hi[:1]


def visibility_decay_loss(P, pts, code_vis):
    """NeRF.visibility_decay_loss (nerf.py:396-414): -logsigmoid(-vis) averaged over random points in the aabb.
    pts (n,3); code_vis (n,32) the instance code of every point's random instance."""
    return -F.logsigmoid(-O.vis_field(P, pts, code_vis)).mean()


def gauss_skin_consistency_loss(P, pts, rest_articulation_mean, code_base_mean):
    """Deformable.gauss_skin_consistency_loss (deformable.py:200-236): weighted BCE pulling the gaussian-bone occupancy towards the
    (detached) field occupancy density / ibeta.  rest_articulation_mean: ((1,B,4),(1,B,4)) = articulation.get_mean_vals();
    code_base_mean (1,32) = the basefield's mean instance code (inst_id=None)."""
    _, centre = O.dual_quaternion_to_quaternion_translation(rest_articulation_mean)
    dist2 = (pts[:, None, :] - centre).pow(2).sum(-1) / (0.01 ** 2)
    density_gauss = (-0.5 * dist2).exp().max(-1)[0][..., None]
    with torch.no_grad():
        density = O.nerf_forward(P, pts[None], {"basefield": code_base_mean}, with_color=False)[0] / P["logibeta"].exp()
        weight_pos = 0.5 / (1e-6 + density.mean())
        weight_neg = 0.5 / (1e-6 + 1 - density).mean()
        weight = density * weight_pos + (1 - density) * weight_neg
    return F.binary_cross_entropy(density_gauss, density, weight=weight)


def soft_deform_loss(P, pts, t_embed, code_fw, code_bw):
    """Deformable.soft_deform_loss + ComposedWarp.compute_post_warp_dist2 (deformable.py:238-252, warping.py:485-503): squared
    displacement of the dense post-warp plus its forward/backward cycle, at random (point, frame, instance) triples.
    pts (n,3); t_embed (n,128) the post-warp time embedding of each point's frame; code_fw / code_bw (n,32)."""
    x = pts[:, None, None]
    x_t = O.dense_warp(P, x, t_embed, code_fw, False)
    dist2 = (x_t - x).pow(2).sum(-1)
    x_back = O.dense_warp(P, x_t, t_embed, code_bw, True)
    return ((dist2 + (x_t - x_back).pow(2).sum(-1)) * 0.5).mean()


def skel_prior_loss(P, prefix, info):
    """ArticulationSkelMLP.skel_prior_loss (pose.py:575-590): rest-pose joint angles and mean log bone-length increments -> 0."""
    so3 = PO.articulation_so3(P, prefix, PO.time_embedding_mean(P, f"{prefix}.time_embedding", info))
    inc = PO.log_bone_len(P, f"{prefix}.log_bone_len", None, 1)
    return so3.pow(2).mean() + 0.02 * inc.pow(2).mean()


def quaternion_translation_to_se3(q, t):
    """quat_transform.py quaternion_translation_to_se3: (M,4),(M,3) -> (M,4,4)."""
    w, x, y, z = q.unbind(-1)
    s = 2.0 / (q * q).sum(-1)
    R = torch.stack([1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w),
                     s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w),
                     s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)], -1).reshape(q.shape[:-1] + (3, 3))
    top = torch.cat([R, t[..., None]], -1)
    bot = torch.tensor([0.0, 0, 0, 1]).expand(q.shape[:-1] + (1, 4))
    return torch.cat([top, bot], -2)


def cam_prior_loss(P, prefix, info, init_vals):
    """CameraMLP.compute_distance_to_prior (time.py:96-105, pose.py:84-90): mse of the (all-frame) SE(3) to the initial cameras."""
    q, t = PO.camera_vals(P, prefix, None, info)
    return F.mse_loss(quaternion_translation_to_se3(q, t), init_vals)
