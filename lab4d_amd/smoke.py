"""One tiny invocation of the hot path on cuda:0 (rays -> warp -> MLP chains -> compositing -> loss -> backward),
checked against the CPU oracle."""
import torch


def run():
    from oracle import lab4d_oracle as O
    from . import deformable as DF
    from . import mlp, synthetic
    M, N, D, res = 2, 4, 8, 64
    P = synthetic.make_weights(3)
    fr = synthetic.add_codes(synthetic.make_frames(4, M, res), P)
    g = torch.Generator().manual_seed(5)
    hxy = torch.cat([torch.rand(M, N, 2, generator=g) * res, torch.ones(M, N, 1)], -1)
    batch = synthetic.make_targets(6, M, N, res, hxy)
    fr["feature"] = batch["feature"]
    rng = {"eik_inds": torch.arange(1), "match_perm": torch.randperm(M * N * D, generator=g)}
    ref = O.render_train(P, fr, hxy, rng, flow_thresh=float(res), n_depth=D)

    Pd = synthetic.to_device(P, "cuda")
    for v in Pd.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    frd = synthetic.add_codes(synthetic.to_device(synthetic.make_frames(4, M, res), "cuda"), Pd)
    bd = synthetic.to_device(batch, "cuda")
    frd["feature"] = bd["feature"]
    out = DF.render_train(Pd, frd, hxy.cuda(), synthetic.to_device(rng, "cuda"), flow_thresh=float(res), n_depth=D, prec=mlp.PREC_F32)
    for k in ("rgb", "mask", "depth", "flow", "feature"):
        a, b = out["rendered"][k].detach().cpu(), ref["rendered"][k].detach()
        err = float((a - b).abs().max() / (b.abs().max() + 1e-12))
        assert err < 2e-4, (k, err)
    loss = sum(DF.losses_fg(out, bd, res, DF.DEFAULT_LOSS_WT).values())
    loss.backward()
    gsum = sum(float(v.grad.abs().sum()) for v in Pd.values() if v.grad is not None)
    assert gsum > 0 and gsum == gsum, "backward produced no / NaN gradients"
    # bf16 MFMA path runs too
    out16 = DF.render_train(Pd, frd, hxy.cuda(), synthetic.to_device(rng, "cuda"), flow_thresh=float(res), n_depth=D, prec=mlp.PREC_BF16)
    mse = float(((out16["rendered"]["rgb"].detach().cpu() - ref["rendered"]["rgb"].detach()) ** 2).mean())
    assert mse < 1e-3, mse
