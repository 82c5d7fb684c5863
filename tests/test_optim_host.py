"""The per-element AdamW arithmetic + segment lookup of csrc/optim.hip (lab4d_amd/csrc/optim_math.hpp), compiled for the
CPU with g++ and compared with torch.optim.AdamW / clip_grad_norm_ on CPU tensors.  CPU only; tests/test_gpu_zoptim.py runs
the kernels."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    out = os.path.join(ROOT, "tests", "host_harness", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "optim_host.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "lab4d_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_harness", "optim_host.cpp"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.adamw_host_step.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_float] * 4 + \
                                   [ctypes.c_int, ctypes.c_float]
    return lib


def test_adamw_rows_match_torch(host):
    g = torch.Generator().manual_seed(0)
    shapes, lrs = [(7, 5), (33,), (4, 4, 3), (1,)], [1e-3, 5e-4, 2e-3, 1e-2]
    params = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    opt = torch.optim.AdamW([{"params": [p], "lr": lr} for p, lr in zip(params, lrs)], betas=(0.9, 0.999), weight_decay=1e-4, foreach=False)
    offs, ends, off = [], [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + 3) // 4 * 4
        ends.append(off)
    flat, m, v = np.zeros(off, np.float32), np.zeros(off, np.float32), np.zeros(off, np.float32)
    for p, o in zip(params, offs):
        flat[o:o + p.numel()] = p.detach().numpy().ravel()
    seg_end, seg_lr = np.asarray(ends, np.int64), np.asarray(lrs, np.float32)
    for step in range(1, 6):
        grads = [torch.randn(s, generator=g) * (10.0 if step == 3 else 1.0) for s in shapes]
        for p, gr in zip(params, grads):
            p.grad = gr.clone()
        norm = torch.nn.utils.clip_grad_norm_(params, 5.0)  # scales p.grad in place
        coef = min(1.0, 5.0 / (float(norm) + 1e-6))
        opt.step()
        fg = np.zeros(off, np.float32)
        for gr, o in zip(grads, offs):
            fg[o:o + gr.numel()] = gr.numpy().ravel()
        host.adamw_host_step(flat.ctypes.data, fg.ctypes.data, m.ctypes.data, v.ctypes.data, off, seg_end.ctypes.data, seg_lr.ctypes.data, len(ends),
                             0.9, 0.999, 1e-8, 1e-4, step, coef)
        for p, o in zip(params, offs):
            ref = p.detach().numpy().ravel()
            assert np.allclose(flat[o:o + p.numel()], ref, rtol=2e-6, atol=1e-7), (step, o)


def test_flat_adamw_host_logic(host, monkeypatch):
    """lab4d_amd.optim.FlatAdamW's host logic (flat layout, p.data / p.grad views, per-parameter rates, clip coefficient, version
    bump) with the library calls routed to the CPU build of the same arithmetic -- a test double, CPU tensors never reach the
    product's library."""
    from lab4d_amd import _lib, optim

    class FakeLib:
        @staticmethod
        def lab4d_grad_norm_clip(g, n, max_norm, work, norm, coef, stream):
            host.grad_norm_clip_host.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
            return host.grad_norm_clip_host(g, n, max_norm, norm, coef)

        @staticmethod
        def lab4d_adamw_step(p, g, m, v, n, seg_end, seg_lr, nseg, b1, b2, eps, wd, step, scale, stream):
            s = 1.0 if scale is None else ctypes.cast(scale, ctypes.POINTER(ctypes.c_float))[0]
            return host.adamw_host_step(p, g, m, v, n, seg_end, seg_lr, nseg, b1, b2, eps, wd, step, s)

    monkeypatch.setattr(_lib, "lib", lambda: FakeLib)
    monkeypatch.setattr(_lib, "require_device", lambda *a: None)
    monkeypatch.setattr(_lib, "stream", lambda: None)
    g = torch.Generator().manual_seed(1)
    shapes, lrs = [(6, 3), (5,), (2, 2, 2)], [1e-3, 3e-3, 5e-4]
    init = [torch.randn(s, generator=g) for s in shapes]
    ref = [x.clone().requires_grad_(True) for x in init]
    mine = [x.clone().requires_grad_(True) for x in init]
    ropt = torch.optim.AdamW([{"params": [p], "lr": lr} for p, lr in zip(ref, lrs)], betas=(0.9, 0.999), weight_decay=1e-4)
    opt = optim.FlatAdamW(mine, lrs)
    assert opt.n == 20 + 8 + 8 and all(p.data_ptr() == opt.flat.data_ptr() + 4 * o for p, o in zip(mine, opt.offsets))
    ws = [torch.randn(s, generator=g) for s in shapes]
    for step in range(4):
        scale = 30.0 if step == 2 else 1.0
        for ps, o in ((ref, ropt), (mine, opt)):
            o.zero_grad()
            (sum((p * p * w).sum() for p, w in zip(ps, ws)) * scale).backward()
        v0 = mine[0]._version
        tn = torch.nn.utils.clip_grad_norm_(ref, 5.0)
        ropt.step()
        opt.step(max_norm=5.0)
        assert abs(float(opt.norm) - float(tn)) <= 1e-5 * float(tn)
        assert mine[0]._version > v0
        for a, b in zip(mine, ref):
            assert torch.allclose(a, b, rtol=3e-6, atol=1e-7), step
    opt.set_lr([0.0, 0.0, 0.0])
    before = [p.detach().clone() for p in mine]
    opt.step()
    assert all(torch.equal(a, b) for a, b in zip(mine, before))  # lr = 0: decay and update both vanish
