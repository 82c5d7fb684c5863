"""sample_pdf's per-ray arithmetic (lab4d_amd/csrc/sample_pdf_math.hpp, the body of k_sample_pdf) compiled for the CPU with g++ and held BIT FOR
BIT to the reference's own function (lab4d/utils/render_utils.py:187-233 restated in oracle.lab4d_oracle.sample_pdf: torch CPU kernels) --
searchsorted indices and samples, the u = 1 end point included.  That end point depends on the last ulp of the row normaliser
torch.sum(weights + eps, -1), i.e. on the ORDER in which ATen's CPU kernel adds the row up (4 vector accumulators of 8 lanes, then lanes
and tail sequentially); the header reproduces that order.  CPU only; tests/test_gpu_ops.py runs the same header on the device."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import lab4d_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    out = os.path.join(ROOT, "tests", "host_harness", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "sample_pdf_host.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "lab4d_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_harness", "sample_pdf_host.cpp"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.row_sum_host.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float]
    lib.row_sum_host.restype = ctypes.c_float
    lib.sample_pdf_host.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_float] + [ctypes.c_void_p] * 2
    return lib


def rows(seed, R, n):
    g = torch.Generator().manual_seed(seed)
    w = torch.rand(R, n, generator=g) ** 3  # compositing weights: most of the mass in a few bins
    w[::5, n // 4: n // 2] = 0              # runs of empty bins
    return w.contiguous()


@pytest.mark.parametrize("n", [1, 3, 7, 8, 14, 30, 31, 62, 63, 126, 254, 511, 512, 700, 2100, 9000])
def test_row_normaliser_equals_torch_sum_bit_for_bit(host, n):
    """torch.sum(weights + eps, -1) for every row length the renderer uses (n_depth // 2 - 2 = 14, 30, 62, 126) and the lengths that reach
    the cascade levels of ATen's kernel (>= 512 elements: 16 rows of 4 vectors)."""
    w = rows(n, 64, n)
    ref = (w + 1e-5).sum(-1)
    a = w.numpy()
    for r in range(w.shape[0]):
        got = host.row_sum_host(a[r].ctypes.data, n, 1e-5)
        assert np.float32(got) == ref[r].numpy(), (n, r, got, float(ref[r]))


def test_linspace_arithmetic(host):
    """The det=True queries: torch.linspace(0, 1, n) on the CPU is start + step * k for the first half and end - step * (n - 1 - k) -- evaluated with a
    FUSED multiply-add by ATen's vectorised kernel -- for the second; the header's formula gives the same floats for every n the renderer can
    use (n_depth // 2 for n_depth = 4 .. 512)."""
    host.linspace01_host.argtypes = [ctypes.c_int, ctypes.c_void_p]
    for n in list(range(2, 260)) + [300, 512, 1000]:
        out = np.empty(n, np.float32)
        host.linspace01_host(n, out.ctypes.data)
        assert np.array_equal(out, torch.linspace(0, 1, n).numpy()), n


def run(host, bins, w, n_imp, u_sorted=None):
    R, n_w = w.shape
    s = np.empty((R, n_imp), np.float32)
    inds = np.empty((R, n_imp), np.int64)
    b, ww = bins.numpy(), w.numpy()
    host.sample_pdf_host(b.ctypes.data, ww.ctypes.data, None if u_sorted is None else u_sorted.numpy().ctypes.data, R, n_w, n_imp, 1e-5, s.ctypes.data,
                         inds.ctypes.data)
    return torch.from_numpy(s), torch.from_numpy(inds)


@pytest.mark.parametrize("n_w,n_imp", [(14, 16), (30, 32), (62, 64), (126, 128)])
def test_indices_and_samples_equal_the_reference_bit_for_bit(host, n_w, n_imp):
    """det=True at the renderer's shapes (n_depth = 32 / 64 / 128 / 256): EVERY index equals torch.searchsorted's on torch's own cdf,
    every sample is the same float."""
    R = 3000
    g = torch.Generator().manual_seed(n_w)
    bins = torch.sort(torch.rand(R, n_w + 1, generator=g), -1)[0].contiguous()
    w = rows(100 + n_w, R, n_w)
    s_ref, i_ref = O.sample_pdf(bins, w, n_imp, return_inds=True)
    s, i = run(host, bins, w, n_imp)
    assert int((i != i_ref).sum()) == 0, int((i != i_ref).sum())
    assert torch.equal(s, s_ref)


def test_same_arithmetic_under_the_avx2_dispatch():
    """ATen registers the AVX2 build of its sum kernel for AVX512 hosts as well (8 float lanes either way): the row sums of a process forced
    to ATEN_CPU_CAPABILITY=avx2 are the same floats as this process's."""
    code = ("import torch; g = torch.Generator().manual_seed(5); w = torch.rand(512, 62, generator=g) ** 3; "
            "print((w + 1e-5).sum(-1).view(torch.int32).sum().item(), torch.backends.cpu.get_cpu_capability())")
    outs = []
    for cap in ("avx2", None):
        env = dict(os.environ)
        if cap:
            env["ATEN_CPU_CAPABILITY"] = cap
        else:
            env.pop("ATEN_CPU_CAPABILITY", None)
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split())
    assert outs[0][0] == outs[1][0], outs
