"""GPU parity of the optimizer kernels (csrc/optim.hip through lab4d_amd.optim.FlatAdamW) against torch.optim.AdamW +
torch.nn.utils.clip_grad_norm_ on the same device tensors.  fp32; the update is element-wise, so the
tolerance is a few ulp of the parameter."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_flat_adamw_matches_torch_adamw():
    from lab4d_amd import optim
    g = torch.Generator().manual_seed(3)
    shapes = [(256, 95), (256,), (3, 128), (1,), (25, 64), (64, 235), (7,)]
    lrs = [5e-4, 5e-4, 1e-3, 1e-2, 2e-4, 5e-4, 1e-3]
    init = [torch.randn(s, generator=g) * 0.2 for s in shapes]
    ref = [x.clone().to(DEV).requires_grad_(True) for x in init]
    mine = [x.clone().to(DEV).requires_grad_(True) for x in init]
    ropt = torch.optim.AdamW([{"params": [p], "lr": lr} for p, lr in zip(ref, lrs)], betas=(0.9, 0.999), weight_decay=1e-4)
    opt = optim.FlatAdamW(mine, lrs)
    ws = [torch.randn(s, generator=g).to(DEV) for s in shapes]
    for step in range(5):
        scale = 40.0 if step in (1, 3) else 1.0  # steps 1 and 3 exceed the clipping threshold
        for ps, o in ((ref, ropt), (mine, opt)):
            o.zero_grad()
            (sum((p * p * w).sum() + (p * w).sum() for p, w in zip(ps, ws)) * scale).backward()
        v0 = mine[0]._version
        tn = torch.nn.utils.clip_grad_norm_(ref, 5.0)
        ropt.step()
        opt.step(max_norm=5.0)
        assert abs(float(opt.norm) - float(tn)) <= 2e-6 * float(tn)
        assert mine[0]._version > v0
        for a, b in zip(mine, ref):
            assert torch.allclose(a, b, rtol=5e-6, atol=2e-7), step
    # unclipped step and a learning-rate change
    opt.set_lr([x * 0.5 for x in lrs])
    for gr, x in zip(ropt.param_groups, lrs):
        gr["lr"] = x * 0.5
    for ps, o in ((ref, ropt), (mine, opt)):
        o.zero_grad()
        sum((p * w).sum() for p, w in zip(ps, ws)).backward()
    ropt.step()
    opt.step()
    for a, b in zip(mine, ref):
        assert torch.allclose(a, b, rtol=5e-6, atol=2e-7)


def test_grad_norm_at_scale():
    """Reduction tree on 16.8 M elements + a ragged tail: the norm against a float64 reference, and coef = min(1, c / (norm + 1e-6))."""
    from lab4d_amd import _lib, optim  # noqa: F401  (optim registers the signatures)
    n = (1 << 24) + 3
    x = torch.randn(n, device=DEV)
    work, norm, coef = torch.empty(512, device=DEV), torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    _lib.check(_lib.lib().lab4d_grad_norm_clip(_lib.ptr(x), n, 5.0, _lib.ptr(work), _lib.ptr(norm), _lib.ptr(coef), _lib.stream()), "grad_norm_clip")
    ref = float(x.double().norm())
    assert abs(float(norm) - ref) <= 1e-5 * ref
    assert abs(float(coef) - min(1.0, 5.0 / (ref + 1e-6))) <= 1e-5
