"""FeatureNeRF.global_match (nnutils/feature.py:152-199) as kernels (csrc/match.hip) against the reference expression in torch, and the
row tap that feeds it the drawn candidates."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    """max |a - b| over max |b| (floored at 5e-2: with one candidate the softmax is constant and the reference gradients are exactly zero, the
    kernel's are rounding residue of 1e-7)."""
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / b.abs().max().clamp_min(5e-2))


@pytest.mark.parametrize("R,K", [(1000, 1024), (4096, 300), (5, 7), (257, 1)])
def test_global_match_equals_the_reference_expression(R, K):
    from lab4d_amd import deformable as DF
    g = torch.Generator().manual_seed(R + K)
    feat_px = torch.nn.functional.normalize(torch.randn(R, 16, generator=g), dim=-1).to(DEV)
    fc0 = torch.nn.functional.normalize(torch.randn(K, 16, generator=g), dim=-1)
    xc0 = torch.randn(K, 3, generator=g) * 0.2
    w = torch.randn(R, 3, generator=g).to(DEV)
    for ls in (0.0, 2.3):  # temperatures exp(logsigma) = 1 (the initial value, feature.py:86) and 10

        def run(fn):
            fc, xc = fc0.to(DEV).requires_grad_(True), xc0.to(DEV).requires_grad_(True)
            logsigma = torch.tensor([ls], device=DEV, requires_grad=True)
            out = fn(feat_px, fc, xc, logsigma)
            return out, torch.autograd.grad((out * w).sum(), [fc, xc, logsigma])

        def reference(f, fc, xc, logsigma):  # feature.py:180-196
            score = torch.matmul(f, fc.t()) * logsigma.exp()
            return torch.sum(torch.softmax(score, dim=1).unsqueeze(-1) * xc, dim=1)

        o1, g1 = run(lambda f, fc, xc, l: DF.GlobalMatch.apply(f, fc, xc, l))
        o0, g0 = run(reference)
        assert rel(o1, o0) < 1e-5, rel(o1, o0)
        for a, b, n in zip(g1, g0, ["feat_c", "xyz_c", "logsigma"]):
            assert rel(a, b) < 2e-5, (n, rel(a, b))


def test_global_match_is_deterministic():
    from lab4d_amd import deformable as DF
    feat_px = torch.randn(3000, 16, device=DEV)
    fc, xc = torch.randn(1024, 16, device=DEV, requires_grad=True), torch.randn(1024, 3, device=DEV, requires_grad=True)
    ls = torch.zeros(1, device=DEV, requires_grad=True)
    res = []
    for _ in range(2):
        out = DF.GlobalMatch.apply(feat_px, fc, xc, ls)
        res.append([out.detach()] + [t.clone() for t in torch.autograd.grad(out.square().sum(), [fc, xc, ls])])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_row_tap_adds_the_rows_gradient_into_the_dense_one():
    from lab4d_amd import deformable as DF
    x = torch.randn(5000, 16, device=DEV, requires_grad=True)
    idx = torch.randperm(5000, device=DEV)[:1024]
    w, v = torch.randn(5000, 16, device=DEV), torch.randn(1024, 16, device=DEV)
    y, rows = DF.RowTap.apply(x, idx)
    assert torch.equal(y, x) and torch.equal(rows, x[idx])
    (gx,) = torch.autograd.grad((y * w).sum() + (y * 2).sum() + (rows * v).sum(), [x])
    ref = (w + 2).index_add(0, idx, v)
    assert torch.allclose(gx, ref, rtol=0, atol=1e-6)
    # only the rows are consumed: the dense gradient is built here
    y, rows = DF.RowTap.apply(x, idx)
    (gx,) = torch.autograd.grad((rows * v).sum(), [x])
    assert torch.allclose(gx, torch.zeros_like(x).index_add(0, idx, v))
