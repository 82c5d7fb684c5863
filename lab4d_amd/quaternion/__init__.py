"""Drop-in for the reference's native `quaternion` package (dqtorch,
lab4d/third_party/quaternion/__init__.py:2-3): same names, same (B,D) calling convention,
double-differentiable autograd chain as in quaternion.py:12-109 and mat3x3.py:13-102 -- backed by
the gfx950 kernels of liblab4d_hip.so instead of the CUDA extension.

A maintainer makes lab4d use it by putting this package first on sys.path or by
`sys.modules["quaternion"] = lab4d_amd.quaternion` before lab4d.utils.quat_transform is
imported (quat_transform.py:10-16)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


def _chk(*ts):
    _lib.require_device(*ts)
    dt = ts[0].dtype
    if dt not in _lib.DTYPE_CODE:
        raise RuntimeError("quaternion ops support float32/float16/float64, got %s" % dt)
    for t in ts:
        if t.dtype != dt:
            raise RuntimeError("quaternion ops: mixed dtypes")
    return _lib.DTYPE_CODE[dt]


class _QuaternionMulBackward(Function):
    @staticmethod
    def forward(ctx, grad, a, b):
        grad, a, b = grad.contiguous(), a.contiguous(), b.contiguous()
        code = _chk(grad, a, b)
        B, D1, D2 = a.shape[0], a.shape[1], b.shape[1]
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        _lib.check(_lib.lib().lab4d_quaternion_mul_backward(_lib.ptr(grad), B, D1, D2, _lib.ptr(a), _lib.ptr(b), _lib.ptr(ga),
                                                            _lib.ptr(gb), code, _lib.stream()), "quaternion_mul_backward")
        ctx.save_for_backward(grad, a, b)
        return ga, gb

    @staticmethod
    @once_differentiable
    def backward(ctx, go1, go2):
        grad, a, b = ctx.saved_tensors
        go1, go2 = go1.contiguous(), go2.contiguous()
        code = _chk(grad, a, b, go1, go2)
        B, D1, D2 = a.shape[0], a.shape[1], b.shape[1]
        gg, gga, ggb = torch.empty_like(grad), torch.empty_like(a), torch.empty_like(b)
        _lib.check(_lib.lib().lab4d_quaternion_mul_backward_backward(
            _lib.ptr(go1), _lib.ptr(go2), B, D1, D2, _lib.ptr(grad), _lib.ptr(a), _lib.ptr(b), _lib.ptr(gg), _lib.ptr(gga),
            _lib.ptr(ggb), code, _lib.stream()), "quaternion_mul_backward_backward")
        return gg, gga, ggb


class _QuaternionMul(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        code = _chk(a, b)
        if a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0] or a.shape[1] not in (3, 4) or b.shape[1] not in (3, 4):
            raise RuntimeError("quaternion_mul expects (B,3|4) x (B,3|4), got %s x %s" % (tuple(a.shape), tuple(b.shape)))
        out = torch.empty(a.shape[0], 4, dtype=a.dtype, device=a.device)
        _lib.check(_lib.lib().lab4d_quaternion_mul_forward(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), a.shape[0], a.shape[1],
                                                           b.shape[1], code, _lib.stream()), "quaternion_mul_forward")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, grad):
        a, b = ctx.saved_tensors
        return _QuaternionMulBackward.apply(grad.contiguous(), a, b)


class _QuaternionConjugate(Function):
    @staticmethod
    def forward(ctx, q):
        q = q.contiguous()
        code = _chk(q)
        out = torch.empty_like(q)
        _lib.check(_lib.lib().lab4d_quaternion_conjugate(_lib.ptr(q), q.numel() // 4, _lib.ptr(out), code, _lib.stream()),
                   "quaternion_conjugate")
        return out

    @staticmethod
    def backward(ctx, grad):
        return _QuaternionConjugate.apply(grad)


quaternion_mul = _QuaternionMul.apply
quaternion_conjugate = _QuaternionConjugate.apply


def mat3x3_det(m):
    x = m.contiguous().view(-1, 9)
    code = _chk(x)
    out = torch.empty(x.shape[0], dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().lab4d_mat3x3_det_forward(_lib.ptr(x), _lib.ptr(out), x.shape[0], code, _lib.stream()), "mat3x3_det")
    return out.view(m.shape[:-2])


def mat3x3_scale_adjoint(m, scales):
    x, s = m.contiguous().view(-1, 9), scales.contiguous().view(-1)
    code = _chk(x, s)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().lab4d_mat3x3_scale_adjoint_forward(_lib.ptr(x), _lib.ptr(s), _lib.ptr(out), x.shape[0], code,
                                                             _lib.stream()), "mat3x3_scale_adjoint")
    return out.view(m.shape)


class _Mat3x3Inv(Function):
    @staticmethod
    def forward(ctx, x):
        code = _chk(x)
        out, scales = torch.empty_like(x), torch.empty(x.shape[0], dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().lab4d_mat3x3_inv_forward(_lib.ptr(x), _lib.ptr(out), _lib.ptr(scales), x.shape[0], code,
                                                       _lib.stream()), "mat3x3_inv_forward")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        (inv,) = ctx.saved_tensors
        grad = grad.contiguous()
        code = _chk(inv, grad)
        gin = torch.empty_like(inv)
        _lib.check(_lib.lib().lab4d_mat3x3_inv_backward(_lib.ptr(grad), _lib.ptr(inv), _lib.ptr(gin), inv.shape[0], code,
                                                        _lib.stream()), "mat3x3_inv_backward")
        return gin


def mat3x3_inv(m):
    return _Mat3x3Inv.apply(m.contiguous().view(-1, 9)).view(m.shape)
