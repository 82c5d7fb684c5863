"""The multiresolution hash encoding's arithmetic (lab4d_amd/csrc/hashgrid_math.hpp, g++ build) against
oracle/hashgrid_oracle.py (an independent vectorised restatement of the Instant-NGP paper; parity vs the reference is
unpinned -- the reference has no hash grid) plus properties the definition implies.  CPU only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import hashgrid_oracle as HO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    out = os.path.join(ROOT, "tests", "host_harness", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "hashgrid_host.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "lab4d_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_harness", "hashgrid_host.cpp"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.hashgrid_host_forward.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    lib.hashgrid_host_backward.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
    return lib


def run_host(host, x, table, res, log2_T, g_out=None):
    S, (L, T, F) = x.shape[0], table.shape
    xa, ta = np.ascontiguousarray(x.numpy(), np.float32), np.ascontiguousarray(table.numpy(), np.float32)
    ra = np.asarray(res, np.int32)
    out = np.empty((S, L * F), np.float32)
    host.hashgrid_host_forward(xa.ctypes.data, ta.ctypes.data, ra.ctypes.data, S, L, log2_T, F, out.ctypes.data)
    if g_out is None:
        return out
    ga = np.ascontiguousarray(g_out.numpy(), np.float32)
    g_table, g_x = np.zeros_like(ta), np.empty_like(xa)
    host.hashgrid_host_backward(xa.ctypes.data, ta.ctypes.data, ra.ctypes.data, ga.ctypes.data, S, L, log2_T, F, g_table.ctypes.data, g_x.ctypes.data)
    return out, g_table, g_x


@pytest.mark.parametrize("L,F,log2_T,n_min,n_max", [(16, 2, 14, 16, 512), (8, 4, 10, 4, 128), (4, 1, 19, 16, 64), (1, 8, 8, 5, 5)])
def test_hash_encode_matches_the_oracle(host, L, F, log2_T, n_min, n_max):
    g = torch.Generator().manual_seed(L * 100 + F)
    res = HO.level_resolutions(L, n_min, n_max)
    assert res[0] == n_min and res[-1] == n_max and all(a <= b for a, b in zip(res, res[1:]))
    S = 257
    x = torch.rand(S, 3, generator=g)
    x[0] = 0.0
    x[1] = 1.0                      # upper boundary: last cell, weight 1
    x[2] = torch.tensor([-0.3, 1.7, 0.5])  # clamped
    table = torch.randn(L, 1 << log2_T, F, generator=g)
    xt, tt = x.clone().requires_grad_(True), table.clone().requires_grad_(True)
    ref = HO.hash_encode(xt, tt, res, log2_T)
    c = torch.randn(S, L * F, generator=g)
    g_x_ref, g_t_ref = torch.autograd.grad((ref * c).sum(), [xt, tt])
    out, g_table, g_x = run_host(host, x, table, res, log2_T, c)
    assert np.allclose(out, ref.detach().numpy(), atol=2e-5)
    assert np.allclose(g_table, g_t_ref.numpy(), atol=2e-5)
    inside = ((x > 0) & (x < 1)).all(-1).numpy()  # the oracle's clamp zeroes d/dx outside [0,1]; the kernel keeps the cell's slope
    assert np.allclose(g_x[inside], g_x_ref.numpy()[inside], atol=2e-3 * max(1.0, float(g_x_ref.abs().max())))


def test_hash_encode_properties(host):
    g = torch.Generator().manual_seed(5)
    L, F, log2_T = 6, 2, 12
    res = HO.level_resolutions(L, 4, 64)
    table = torch.randn(L, 1 << log2_T, F, generator=g)
    # a constant table encodes every point to that constant (the 8 weights sum to one)
    const = torch.full_like(table, 0.75)
    out = run_host(host, torch.rand(64, 3, generator=g), const, res, log2_T)
    assert np.allclose(out, 0.75, atol=1e-6)
    # at a grid vertex of a dense level the encoding is that vertex's table entry
    r = res[0]
    v = torch.tensor([[1, 2, 3], [0, 0, 0], [r, r, r]])
    out = run_host(host, v.float() / r, table, res, log2_T)
    n = r + 1
    idx = v[:, 0] + n * (v[:, 1] + n * v[:, 2])
    assert n ** 3 <= (1 << log2_T)
    assert np.allclose(out[:, :F], table[0][idx].numpy(), atol=1e-6)
    # linear in the table
    x = torch.rand(50, 3, generator=g)
    t2 = torch.randn_like(table)
    a, b, ab = run_host(host, x, table, res, log2_T), run_host(host, x, t2, res, log2_T), run_host(host, x, table * 2 - t2 * 0.5, res, log2_T)
    assert np.allclose(ab, 2 * a - 0.5 * b, atol=1e-5)
    # continuity across a cell face of the hashed (finest) level
    assert (res[-1] + 1) ** 3 > (1 << log2_T)
    xa = torch.tensor([[0.5 - 1e-6, 0.3, 0.7], [0.5 + 1e-6, 0.3, 0.7]])
    o = run_host(host, xa, table, res, log2_T)
    assert np.allclose(o[0], o[1], atol=1e-3)
