"""The fused per-ray loss epilogue (csrc/losses.hip, include/lab4d_loss.h; engine/model.py:401-611) against the same terms written
op by op (deformable.losses_fg_reference_ops, the form the reference itself uses: one masked `v[v > 0].mean()` per term).  The
end-to-end fixtures (tests/test_gpu_field.py) hold the fused form to the reference's own loss values and gradients as well."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(seed, M, N, zeros=True):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.rand(*s, generator=g)
    r = {"mask": rnd(M, N, 1), "rgb": rnd(M, N, 3), "depth": rnd(M, N, 1) * 3, "flow": torch.randn(M, N, 2, generator=g) * 4, "eikonal": rnd(M, N)}
    a = {"feature": torch.nn.functional.normalize(torch.randn(M, N, 16, generator=g), dim=-1), "xy_reproj": rnd(M, N, 2) * 64, "vis": rnd(M, N, 1),
         "gauss_mask": rnd(M, N, 1), "cyc_dist": rnd(M, N, 1) * 0.1, "delta_skin": rnd(M, N) * 0.01, "skin_entropy": rnd(M, N, 1)}
    hxy = torch.cat([rnd(M, N, 2) * 64, torch.ones(M, N, 1)], -1)
    b = {"mask": rnd(M, N, 1) > 0.4, "rgb": rnd(M, N, 3), "depth": torch.ones(M, N, 1), "flow": torch.zeros(M, N, 2), "flow_uct": (rnd(M, N, 1) > 0.2).float(),
         "vis2d": (rnd(M, N, 1) > 0.1).float(), "is_detected": torch.ones(M), "feature": torch.nn.functional.normalize(torch.randn(M, N, 16, generator=g), dim=-1),
         "hxy": hxy}
    if zeros:  # elements that are exactly zero must drop out of both the sum and the count (v > 0), and have zero gradient
        b["is_detected"][1] = 0.0
        r["eikonal"][:, ::3] = 0.0
        a["cyc_dist"][:, 1::4] = 0.0
        r["rgb"][0, :5] = b["rgb"][0, :5]
        r["flow"][0, :7] = b["flow"][0, :7]
        a["feature"][0, :3] = b["feature"][0, :3]
        r["depth"][1, :4] = b["depth"][1, :4]
    return r, a, b


@pytest.mark.parametrize("M,N,seed", [(2, 37, 1), (4, 1000, 2), (2, 16384, 3)])
def test_fused_ray_losses_match_the_op_by_op_form(M, N, seed):
    from lab4d_amd import deformable as DF
    r, a, b = _case(seed, M, N)
    dev = lambda d: {k: v.to(DEV) for k, v in d.items()}
    bd = dev(b)

    def run(fn):
        rd = {k: v.to(DEV).clone().requires_grad_(True) for k, v in r.items()}
        ad = {k: v.to(DEV).clone().requires_grad_(True) for k, v in a.items()}
        L = fn({"rendered": rd, "aux_dict": {"fg": ad}}, bd, 64, DF.DEFAULT_LOSS_WT)
        tot = L.total if getattr(L, "total", None) is not None else sum(L.values())
        leaves = list(rd.values()) + list(ad.values())
        grads = torch.autograd.grad(tot, leaves, allow_unused=True)
        return L, tot, dict(zip(list(rd.keys()) + list(ad.keys()), grads))

    Lf, tf, gf = run(DF.losses_fg)
    Lr, tr, gr = run(DF.losses_fg_reference_ops)
    assert set(Lf.keys()) == set(Lr.keys()) and list(Lf.keys()) == DF.LOSS_TERMS
    for k in Lr:
        assert torch.allclose(Lf[k], Lr[k], rtol=2e-5, atol=1e-9), (k, float(Lf[k]), float(Lr[k]))
    assert torch.allclose(tf, tr, rtol=2e-5)
    assert torch.allclose(tf, sum(Lf.values()), rtol=1e-6)
    for k, g_ref in gr.items():
        assert (g_ref is None) == (gf[k] is None), k
        if g_ref is not None:
            assert torch.allclose(gf[k], g_ref, rtol=1e-4, atol=1e-6 * float(g_ref.abs().max()) + 1e-12), (k, float((gf[k] - g_ref).abs().max()))
    # a per-term upstream gradient (not just the total) is honoured
    rd = {k: v.to(DEV).clone().requires_grad_(True) for k, v in r.items()}
    ad = {k: v.to(DEV).clone().requires_grad_(True) for k, v in a.items()}
    L = DF.losses_fg({"rendered": rd, "aux_dict": {"fg": ad}}, bd, 64, DF.DEFAULT_LOSS_WT)
    (g_rgb,) = torch.autograd.grad(3.0 * L["rgb"] + L["depth"], [rd["rgb"]])
    assert torch.allclose(g_rgb, 3.0 * gr["rgb"] * 0 + 3.0 * torch.autograd.grad(DF.losses_fg_reference_ops(
        {"rendered": {k: (v if k != "rgb" else rd["rgb"]) for k, v in rd.items()}, "aux_dict": {"fg": ad}}, bd, 64, DF.DEFAULT_LOSS_WT)["rgb"], [rd["rgb"]])[0],
        rtol=1e-4, atol=1e-9)


def test_precomputed_balance_weights_are_used():
    from lab4d_amd import deformable as DF
    r, a, b = _case(5, 2, 64, zeros=False)
    dev = lambda d: {k: v.to(DEV) for k, v in d.items()}
    res = {"rendered": dev(r), "aux_dict": {"fg": dev(a)}}
    bd = dev(b)
    L0 = DF.losses_fg(res, bd, 64, None)
    bd2 = dict(bd, mask_balance_wt=DF.mask_balance_wt(bd["mask"], bd["vis2d"], bd["is_detected"]))
    L1 = DF.losses_fg(res, bd2, 64, None)
    assert torch.equal(L0["mask"], L1["mask"])
    bd3 = dict(bd, mask_balance_wt=2 * bd2["mask_balance_wt"])
    assert torch.allclose(DF.losses_fg(res, bd3, 64, None)["mask"], 2 * L0["mask"], rtol=1e-6)


@pytest.mark.parametrize("M,N,seed", [(2, 37, 11), (2, 16384, 12)])
def test_fused_comp_losses_match_the_op_by_op_form(M, N, seed):
    """field_type "comp" (engine/model.py:455-461, 486-493, 566-571) through the same kernels: foreground mask + opaque composite,
    background visibility at 1 %, dense terms masked by vis2d only -- values and every input gradient against the op-by-op form."""
    from lab4d_amd import deformable as DF
    r, a, b = _case(seed, M, N)
    g = torch.Generator().manual_seed(seed + 100)
    r["mask_fg"] = torch.rand(M, N, 1, generator=g)
    bg = {"vis": torch.rand(M, N, 1, generator=g)}
    bd = {k: v.to(DEV) for k, v in b.items()}

    def run(fn):
        rd = {k: v.to(DEV).clone().requires_grad_(True) for k, v in r.items()}
        ad = {k: v.to(DEV).clone().requires_grad_(True) for k, v in a.items()}
        gd = {k: v.to(DEV).clone().requires_grad_(True) for k, v in bg.items()}
        L = fn({"rendered": rd, "aux_dict": {"fg": ad, "bg": gd}}, bd, 64, DF.DEFAULT_LOSS_WT)
        tot = L.total if getattr(L, "total", None) is not None else sum(L.values())
        names = ["r." + k for k in rd] + ["a." + k for k in ad] + ["b." + k for k in gd]
        grads = torch.autograd.grad(tot, list(rd.values()) + list(ad.values()) + list(gd.values()), allow_unused=True)
        return L, tot, dict(zip(names, grads))

    Lf, tf, gf = run(DF.losses_comp)
    Lr, tr, gr = run(DF.losses_comp_reference_ops)
    assert set(Lf.keys()) == set(Lr.keys())
    for k in Lr:
        assert torch.allclose(Lf[k], Lr[k], rtol=2e-5, atol=1e-9), (k, float(Lf[k]), float(Lr[k]))
    assert torch.allclose(tf, tr, rtol=2e-5)
    for k, g_ref in gr.items():
        assert (g_ref is None) == (gf[k] is None), k
        if g_ref is not None:
            assert torch.allclose(gf[k], g_ref, rtol=1e-4, atol=1e-6 * float(g_ref.abs().max()) + 1e-12), (k, float((gf[k] - g_ref).abs().max()))
