"""Host-side mirror of lab4d/nnutils/multifields.py MultiFields.compose_fields (lines 339-398) on the gfx950 kernels of
csrc/compose.hip: per-ray z-merge of the samples of two fields (fg + bg, "comp" configs)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib

vp, ci = _lib.vp, _lib.ci
_lib.register("lab4d_compose_order", [vp, ci, vp, ci, ci, vp, vp, vp])
_lib.register("lab4d_compose_gather", [vp, ci, vp, ci, vp, ci, ci, ci, ci, vp, vp])


def compose_order(depth_a, depth_b):
    """argsort of cat([depth_a, depth_b], 2) per ray and its inverse: (M,N,Da,1),(M,N,Db,1) -> int32 (R,Da+Db) x2."""
    da, db = depth_a.detach().contiguous().float(), depth_b.detach().contiguous().float()
    _lib.require_device(da, db)
    Da, Db = da.shape[2], db.shape[2]
    R = da.shape[0] * da.shape[1]
    order = torch.empty(R, Da + Db, dtype=torch.int32, device=da.device)
    pos = torch.empty_like(order)
    _lib.check(_lib.lib().lab4d_compose_order(_lib.ptr(da), Da, _lib.ptr(db), Db, R, _lib.ptr(order), _lib.ptr(pos), _lib.stream()), "compose_order")
    return order, pos


class _ComposeGather(Function):
    """out (M,N,Da+Db,C) = cat([a, b], 2) gathered with `order`; a or b may be None (zeros)."""

    @staticmethod
    def forward(ctx, a, b, order, pos, Da, Db):
        ref = a if a is not None else b
        M, N, _, C = ref.shape
        R = M * N
        a_c = a.contiguous().float() if a is not None else None
        b_c = b.contiguous().float() if b is not None else None
        out = torch.empty(M, N, Da + Db, C, device=ref.device)
        _lib.check(_lib.lib().lab4d_compose_gather(_lib.ptr(a_c), Da, _lib.ptr(b_c), Db, _lib.ptr(order), Da + Db, R, Da + Db, C, _lib.ptr(out),
                                                   _lib.stream()), "compose_gather")
        ctx.save_for_backward(pos)
        ctx.meta = (M, N, Da, Db, C, a is not None, b is not None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (pos,) = ctx.saved_tensors
        M, N, Da, Db, C, has_a, has_b = ctx.meta
        R, Dt = M * N, Da + Db
        g = g.contiguous()
        ga = gb = None
        if has_a and ctx.needs_input_grad[0]:
            ga = torch.empty(M, N, Da, C, device=g.device)
            _lib.check(_lib.lib().lab4d_compose_gather(_lib.ptr(g), Dt, None, 0, _lib.ptr(pos), Dt, R, Da, C, _lib.ptr(ga), _lib.stream()),
                       "compose_gather(adjoint a)")
        if has_b and ctx.needs_input_grad[1]:
            gb = torch.empty(M, N, Db, C, device=g.device)
            pos_b = pos[:, Da:]  # row stride stays Dt: pass the offset view's pointer
            _lib.check(_lib.lib().lab4d_compose_gather(_lib.ptr(g), Dt, None, 0, _lib.ptr_at(pos_b), Dt, R, Db, C, _lib.ptr(gb), _lib.stream()),
                       "compose_gather(adjoint b)")
        return ga, gb, None, None, None, None


def compose_fields(multifields_dict, deltas_dict):
    """MultiFields.compose_fields (multifields.py:339-398) for one or two fields.  Same contract as the reference:
    ({cat: {key: (M,N,D_cat,c)}}, {cat: (M,N,D_cat,1)}) -> ({key: (M,N,sum D,c)}, deltas (M,N,sum D,1)).  Keys one field
    does not produce are zero-filled; samples are z-sorted by the concatenated "depth" (stable on ties)."""
    cats = list(multifields_dict.keys())
    if len(cats) == 1:
        return multifields_dict[cats[0]], deltas_dict[cats[0]]
    if len(cats) != 2:
        raise RuntimeError("compose_fields: one or two fields supported, got %d" % len(cats))
    fa, fb = multifields_dict[cats[0]], multifields_dict[cats[1]]
    Da, Db = fa["depth"].shape[2], fb["depth"].shape[2]
    order, pos = compose_order(fa["depth"], fb["depth"])
    keys = list(fa.keys()) + [k for k in fb.keys() if k not in fa]
    out = {k: _ComposeGather.apply(fa.get(k), fb.get(k), order, pos, Da, Db) for k in keys}
    deltas = _ComposeGather.apply(deltas_dict[cats[0]], deltas_dict[cats[1]], order, pos, Da, Db)
    return out, deltas
