"""Per-FRAME quaternion / dual-quaternion algebra in plain torch (device tensors of shape (M,4),
(M,B,4)): a few hundred elements per step, outside the per-sample hot path.  Mirrors the helpers of
lab4d/utils/quat_transform.py that the field code applies to per-frame quantities; anything
per-sample goes through the HIP kernels instead."""
import torch

from . import quaternion as _hipq


def _pad_w(a):
    if a.shape[-1] == 3:
        a = torch.cat([torch.zeros_like(a[..., :1]), a], -1)
    return a


def quaternion_mul(a, b):
    """quat_transform.py:62-81 / 106-113: on device tensors this is ONE launch of the dqtorch-replacement kernel
    (double-differentiable), exactly like the reference dispatches to its CUDA kernel; a 3-vector operand is a pure
    quaternion (quaternion.cu:46-57)."""
    if a.is_cuda and a.shape[:-1] == b.shape[:-1]:
        out_shape = a.shape[:-1] + (4,)
        return _hipq.quaternion_mul(a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])).view(out_shape)
    a, b = torch.broadcast_tensors(_pad_w(a), _pad_w(b))
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


def quaternion_conjugate(q):
    if q.is_cuda:
        return _hipq.quaternion_conjugate(q.reshape(-1, 4)).view(q.shape)
    return torch.cat((q[..., :1], -q[..., 1:]), -1)


def quaternion_apply(q, p):
    return quaternion_mul(quaternion_mul(q, p), quaternion_conjugate(q))[..., 1:]


def quaternion_translation_inverse(q, t):
    """quat_transform.py:282-287."""
    qi = quaternion_conjugate(q)
    return qi, quaternion_apply(qi, -t)


def dual_quaternion_mul(dq1, dq2):
    """quat_transform.py:430-438."""
    return quaternion_mul(dq1[0], dq2[0]), quaternion_mul(dq1[0], dq2[1]) + quaternion_mul(dq1[1], dq2[0])


def dual_quaternion_inverse(dq):
    return quaternion_conjugate(dq[0]), quaternion_conjugate(dq[1])


def dual_quaternion_to_quaternion_translation(dq):
    """quat_transform.py:337-344."""
    return dq[0], 2 * quaternion_mul(dq[1], quaternion_conjugate(dq[0]))[..., 1:]


def kmatinv(K):
    """geom_utils.Kmatinv (geom_utils.py:308-341) for pinhole matrices [[fx,0,px],[0,fy,py],[0,0,1]]."""
    fx, fy, px, py = K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2]
    z, o = torch.zeros_like(fx), torch.ones_like(fx)
    return torch.stack([torch.stack([1 / fx, z, -px / fx], -1), torch.stack([z, 1 / fy, -py / fy], -1),
                        torch.stack([z, z, o], -1)], -2)
