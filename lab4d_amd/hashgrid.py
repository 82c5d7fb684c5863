"""Multiresolution hash encoding (Mueller et al. 2022) on the gfx950 kernels of csrc/hashgrid.hip -- BASELINE config 5's
encoding variant.  The reference has no such module (lab4d/nnutils/nerf.py:98 is a TODO), so this mirrors no reference
interface; it is shaped like `PosEmbedding.forward` (embedding.py:69-125: (..., 3) -> (..., C)) so that it can stand in front of
a basefield.  Parity is unpinned against the reference; the arithmetic is checked against oracle/hashgrid_oracle.py.
"""
import math

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib

vp, ci = _lib.vp, _lib.ci
_lib.register("lab4d_hashgrid_forward", [vp, vp, vp, ci, ci, ci, ci, vp, vp])
_lib.register("lab4d_hashgrid_forward_inside", [vp, vp, vp, ci, ci, ci, ci, vp, vp])
_lib.register("lab4d_hashgrid_backward", [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp])
_lib.register("lab4d_hashgrid_absmax", [vp, __import__("ctypes").c_long, vp, vp])
_lib.register("lab4d_hashgrid_backward_f16", [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp])
_lib.register("lab4d_hashgrid_flush_f16", [vp, vp, ci, ci, ci, vp, vp])


def level_resolutions(L, n_min, n_max):
    """N_l = floor(N_min * b^l), b = exp((ln N_max - ln N_min) / (L - 1))  (paper eq. 2-3)."""
    b = math.exp((math.log(n_max) - math.log(n_min)) / (L - 1)) if L > 1 else 1.0
    return [int(math.floor(n_min * b ** l + 1e-9)) for l in range(L)]


_G16 = {}


def first_hashed_level(res_list, log2_T):
    """Index of the first level whose (res + 1)^3 vertices do not fit the table (hashgrid_math.hpp vertex_index): the levels in front of it are
    direct-indexed (dense), it and the finer ones go through the spatial hash."""
    for l, r in enumerate(res_list):
        if (int(r) + 1) ** 3 > (1 << log2_T):
            return l
    return len(res_list)


class _HashEncode(Function):
    @staticmethod
    def forward(ctx, x, table, res, log2_T, inside_only=False, f16_from=None):
        x, table = x.contiguous().float(), table.contiguous().float()
        _lib.require_device(x, table, res)
        S, (L, T, F) = x.shape[0], table.shape
        if T != 1 << log2_T or res.numel() != L:
            raise RuntimeError("hash_encode: table must be (L, 2^log2_T, F) with one resolution per level")
        out = torch.empty(S, L * F, device=x.device)
        # algorithmic bytes: 8 vertices x F floats gathered per level, the point read, L*F floats written
        with _lib.timed("k_hashgrid_fwd", (0.0, 4.0 * S * (8 * L * F + 3 + L * F))):
            fn = _lib.lib().lab4d_hashgrid_forward_inside if inside_only else _lib.lib().lab4d_hashgrid_forward
            _lib.check(fn(_lib.ptr(x), _lib.ptr(table), _lib.ptr(res), S, L, log2_T, F, _lib.ptr(out), _lib.stream()), "hashgrid_forward")
        ctx.save_for_backward(x, table, res)
        ctx.log2_T, ctx.f16_from = log2_T, f16_from
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, table, res = ctx.saved_tensors
        S, (L, T, F) = x.shape[0], table.shape
        g = g.contiguous().float()
        g_table = torch.zeros_like(table) if ctx.needs_input_grad[1] else None
        g_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        if g_table is None and g_x is None:
            return None, None, None, None, None, None
        if g_table is not None and ctx.f16_from is not None and ctx.f16_from < L:
            # the hashed levels' table gradient through packed 2 x fp16 atomics at a per-launch power-of-two scale (include/lab4d_hashgrid.h)
            if F != 2:
                raise NotImplementedError("hash_encode: the packed fp16 table gradient needs F = 2 (two features per vertex = one 32-bit word)")
            key = (x.device, L, ctx.log2_T)
            if key not in _G16:  # persistent scratch (one per device and table shape; launches on ONE stream at a time): the flush kernel hands the words back cleared, so a launch costs no 33 MB zero-fill.  Create it OUTSIDE a hipGraph capture (bench.py's eager warm-up does): a buffer born in a graph's private pool must not be touched by eager calls.
                _G16[key] = (torch.zeros(L << ctx.log2_T, dtype=torch.int32, device=x.device), torch.zeros(1, dtype=torch.int32, device=x.device))
            g16, amax = _G16[key]
            amax.zero_()
            lib = _lib.lib()
            with _lib.timed("k_hashgrid_bwd", (0.0, 4.0 * S * (2 * 8 * L * F + 6 + L * F))):
                _lib.check(lib.lab4d_hashgrid_absmax(_lib.ptr(g), g.numel(), _lib.ptr(amax), _lib.stream()), "hashgrid_absmax")
                _lib.check(lib.lab4d_hashgrid_backward_f16(_lib.ptr(x), _lib.ptr(table), _lib.ptr(res), _lib.ptr(g), S, L, ctx.log2_T, int(ctx.f16_from),
                                                           _lib.ptr(g_table), _lib.ptr(g16), _lib.ptr(amax), _lib.ptr(g_x), _lib.stream()), "hashgrid_backward_f16")
                _lib.check(lib.lab4d_hashgrid_flush_f16(_lib.ptr(g16), _lib.ptr(amax), L, ctx.log2_T, int(ctx.f16_from), _lib.ptr(g_table), _lib.stream()),
                           "hashgrid_flush_f16")
            return g_x, g_table, None, None, None, None
        with _lib.timed("k_hashgrid_bwd", (0.0, 4.0 * S * (2 * 8 * L * F + 6 + L * F))):  # vertices read (d/dx) and atomically added to
            _lib.check(_lib.lib().lab4d_hashgrid_backward(_lib.ptr(x), _lib.ptr(table), _lib.ptr(res), _lib.ptr(g), S, L, ctx.log2_T, F, _lib.ptr(g_table),
                                                          _lib.ptr(g_x), _lib.stream()), "hashgrid_backward")
        return g_x, g_table, None, None, None, None


def hash_encode(x, table, res, log2_T, inside_only=False, f16_from=None):
    """x (..., 3) in [0,1]^3, table (L, 2^log2_T, F), res: int32 device tensor (L) from level_resolutions -> (..., L*F).
    inside_only: points outside [0,1]^3 get the zero encoding and never touch the table (a field defined on the box masks them anyway).
    f16_from: None = the table gradient through fp32 atomics (exact accumulation); an int = the levels from that index on (use first_hashed_level)
    accumulate theirs through packed 2 x fp16 atomics at a per-launch scale (F = 2): half the atomics, fp16 rounding of every run's contribution."""
    out = _HashEncode.apply(x.reshape(-1, 3), table, res, log2_T, inside_only, f16_from)
    return out.view(x.shape[:-1] + (out.shape[-1],))
