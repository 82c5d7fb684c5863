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


@pytest.mark.parametrize("prec_name", ["f32", "bf16"])
def test_fused_gradient_accumulation_equals_autograd(golden_dir, prec_name):
    """mlp.FUSED_GRAD_ACCUM (the training loop's mode: weight-gradient kernels add straight into the optimizer's flat gradient
    views, in reference layout, lab4d_mlp_wgrad_mapped) gives the same gradients as handing them to autograd -- over the whole
    training graph of the reference fixture, two accumulation passes (gradient accumulation over chunks), every parameter."""
    import os
    from lab4d_amd import deformable as DF
    from lab4d_amd import mlp, optim, synthetic
    prec = mlp.PREC_F32 if prec_name == "f32" else mlp.PREC_BF16
    g = torch.load(os.path.join(golden_dir, "train_small.pt"), weights_only=False)
    meta = g["meta"]

    def run(fused):
        P = synthetic.to_device(synthetic.make_weights(meta["seed"]), DEV)
        for k, v in P.items():
            if v.dtype.is_floating_point and k != "aabb":
                v.requires_grad_(True)
        params = [v for v in P.values() if v.requires_grad]
        opt = optim.FlatAdamW(params, 5e-4)
        opt.zero_grad()
        mlp.FUSED_GRAD_ACCUM = fused
        try:
            for _ in range(2):
                fr = synthetic.add_codes(synthetic.to_device(dict(g["frames"]), DEV), P)
                batch = synthetic.to_device(g["batch"], DEV)
                fr["feature"] = batch["feature"]
                res = DF.render_train(P, fr, g["hxy"].to(DEV), synthetic.to_device(g["rng"], DEV), flow_thresh=meta["flow_thresh"], n_depth=meta["D"],
                                      alpha=meta["alpha"], prec=prec)
                sum(DF.losses_fg(res, batch, meta["res"], DF.DEFAULT_LOSS_WT).values()).backward()
        finally:
            mlp.FUSED_GRAD_ACCUM = False
        return {k: v.grad.clone() for k, v in P.items() if v.requires_grad}, opt.flat_grad.clone()

    ga, fa = run(False)
    gb, fb = run(True)
    worst = 0.0
    for k in ga:
        denom = float(ga[k].abs().max()) + 1e-20
        e = float((ga[k] - gb[k]).abs().max()) / denom
        worst = max(worst, e)
        assert e < 2e-5, (k, e)  # same kernels, same operands: only the order of the fp32 atomic adds differs
    assert float(fa.abs().sum()) > 0 and torch.allclose(fa, fb, rtol=1e-3, atol=1e-6 * float(fa.abs().max()))


def test_repack_all_refreshes_packed_weights_in_place():
    """After an optimizer step the packed bf16 copies a captured hipGraph reads are refreshed IN PLACE (same device address):
    mlp.repack_all() re-packs exactly the copies whose parameter changed, and a forward pass then sees the new weights."""
    from lab4d_amd import mlp, synthetic
    P = synthetic.to_device(synthetic.make_weights(5), DEV)
    fr = synthetic.add_codes(synthetic.to_device(synthetic.make_frames(6, 2, 64), DEV), P)
    x = (torch.rand(512, 3, device=DEV) - 0.5) * 0.2
    mlp.clear_caches()
    with torch.no_grad():
        y0 = mlp.run_chain(mlp.NET_VIS, mlp.PREC_BF16, P, x, 256, conds={0: fr["code_vis"]}).clone()
        W = P["vis_mlp.basefield.linear_2.0.weight"]
        buf = mlp.packed_weights(mlp.NET_VIS, 1, mlp.PREC_BF16, W, False)
        addr, before = buf.data_ptr(), buf.clone()
        assert mlp.repack_all() == 0  # nothing changed yet
        W.mul_(1.5)  # in-place update bumps the version counter, like an optimizer step
        n = mlp.repack_all()
        assert n >= 1
        buf2 = mlp.packed_weights(mlp.NET_VIS, 1, mlp.PREC_BF16, W, False)
        assert buf2.data_ptr() == addr and not torch.equal(buf2, before)
        y1 = mlp.run_chain(mlp.NET_VIS, mlp.PREC_BF16, P, x, 256, conds={0: fr["code_vis"]})
        assert float((y1 - y0).abs().max()) > 1e-4


def test_adopted_optimizer_steps_with_the_scheduler_attached():
    """The reference's order (engine/trainer.py:185-207): AdamW, THEN OneCycleLR on it -- LRScheduler.__init__ leaves an instance attribute
    opt.step wrapping the bound AdamW.step, which survives a class swap.  After TorchFlatAdamW.adopt() the training loop's
    check_grad(); optimizer.step(); scheduler.step() must run the flat kernel: weights move, the device step count advances, the next
    step uploads the scheduler's new rates."""
    from lab4d_amd.optim import TorchFlatAdamW
    g = torch.Generator().manual_seed(4)
    ps = [torch.nn.Parameter((torch.randn(s, generator=g) * 0.2).to(DEV)) for s in [(64, 95), (64,), (3, 64), (1,)]]
    ref = [p.detach().clone().requires_grad_(True) for p in ps]
    groups = lambda qs: [{"params": [p], "lr": lr} for p, lr in zip(qs, (5e-4, 5e-4, 1e-3, 5e-3))]  # noqa: E731
    mk = lambda qs: torch.optim.AdamW(groups(qs), betas=(0.9, 0.999), weight_decay=1e-4)  # noqa: E731
    sched = lambda o: torch.optim.lr_scheduler.OneCycleLR(o, [5e-4, 5e-4, 1e-3, 5e-3], total_steps=10, pct_start=0.3, cycle_momentum=False)  # noqa: E731
    opt, ropt = mk(ps), mk(ref)
    sch, rsch = sched(opt), sched(ropt)
    assert "step" in opt.__dict__  # the stale wrapper the scheduler installed
    TorchFlatAdamW.adopt(opt)
    assert isinstance(opt, TorchFlatAdamW) and sch.optimizer is opt
    ws = [torch.randn(p.shape, generator=g).to(DEV) for p in ps]
    for it in range(3):
        for qs, o in ((ps, opt), (ref, ropt)):
            o.zero_grad()
            (sum((p * p * w).sum() + (p * w).sum() for p, w in zip(qs, ws)) * 0.02).backward()  # norm ~ 2: below the discard threshold
        opt.check_grad(5.0)
        tn = torch.nn.utils.clip_grad_norm_(ref, 5.0)
        opt.step()
        sch.step()
        ropt.step()
        rsch.step()
        assert abs(float(opt.flat.norm) - float(tn)) <= 2e-6 * float(tn)
        assert int(opt.flat.dev_step) == it + 1 and int(opt.skipped) == 0
        for a, b in zip(ps, ref):
            assert torch.allclose(a, b, rtol=5e-6, atol=2e-7), it
    assert getattr(opt, "_opt_called", False)  # the scheduler's call-order bookkeeping still works
