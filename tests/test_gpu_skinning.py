"""GPU parity of the skinning warp (bone coords -> delta-skin MLP -> dual-quaternion blend) vs the oracle."""
import pytest
import torch

from lab4d_amd import synthetic
from oracle import lab4d_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("backward", [True, False])
@pytest.mark.parametrize("shape", [(2, 7, 9), (4, 3, 16)])
def test_skinning_warp_forward_backward(backward, shape):
    from lab4d_amd import warping
    M, N, D = shape
    P = synthetic.make_weights(4)
    fr = synthetic.add_codes(synthetic.make_frames(5, M, 64), P)
    g = torch.Generator().manual_seed(6)
    xyz = torch.randn(M, N, D, 3, generator=g) * 0.08
    w_out = torch.randn(M, N, D, 3, generator=g)
    w_e = torch.randn(M, N, D, 1, generator=g)
    w_d = torch.randn(M, N, D, 1, generator=g) * 100
    pkeys = ["warp.skinning_model.log_gauss", "warp.skinning_model.delta_field.linear_1.0.weight",
             "warp.skinning_model.delta_field.linear_final.weight", "warp.skinning_model.delta_field.linear_2.0.bias"]

    def run(dev, fn):
        Pl = {k: (v.to(dev).clone().requires_grad_(True) if v.dtype.is_floating_point else v.to(dev)) for k, v in P.items()}
        leaves = {}
        frl = {}
        for k, v in fr.items():
            if isinstance(v, tuple):
                frl[k] = tuple(t.to(dev).clone().requires_grad_(True) for t in v)
                for i, t in enumerate(frl[k]):
                    leaves[f"{k}.{i}"] = t
            elif v.dtype.is_floating_point:
                frl[k] = v.to(dev).clone().requires_grad_(True)
                leaves[k] = frl[k]
            else:
                frl[k] = v.to(dev)
        x = xyz.to(dev).clone().requires_grad_(True)
        te = frl["t_embed"] if backward else frl["t_embed_mean"]
        out, aux = fn(Pl, x, frl["t_articulation"], frl["rest_articulation"], te, frl["code_skin"], backward)
        loss = (out * w_out.to(dev)).sum() + (aux["skin_entropy"] * w_e.to(dev)).sum() + (aux["delta_skin"] * w_d.to(dev)).sum()
        names = ["t_articulation.0", "t_articulation.1", "rest_articulation.0", "rest_articulation.1", "code_skin",
                 "t_embed" if backward else "t_embed_mean"]
        gs = torch.autograd.grad(loss, [x] + [Pl[k] for k in pkeys] + [leaves[k] for k in names])
        return (out, aux["skin_entropy"], aux["delta_skin"]), dict(zip(["x"] + pkeys + names, gs))

    ro, rg = run("cpu", O.skinning_warp)
    do, dg = run(DEV, warping.skinning_warp)
    for a, b, n in zip(do, ro, ["xyz", "skin_entropy", "delta_skin"]):
        assert rel(a, b) < 1e-4, f"{n}: {rel(a, b):.3e}"
    for k in rg:
        assert rel(dg[k], rg[k]) < 1e-3, f"grad {k}: {rel(dg[k], rg[k]):.3e}"


@pytest.mark.parametrize("backward", [True, False])
def test_composed_warp_dense_post_warp(backward):
    """ComposedWarp (skinning + DenseWarp post-warp, LAB4D_NET_DENSE) forward + all gradients vs the oracle, fp32."""
    from lab4d_amd import warping
    M, N, D = 2, 9, 11
    P = synthetic.add_dense_weights(synthetic.make_weights(4), seed=4)
    fr = synthetic.add_codes(synthetic.make_frames(5, M, 64), P)
    g = torch.Generator().manual_seed(8)
    xyz = torch.randn(M, N, D, 3, generator=g) * 0.08
    w_out = torch.randn(M, N, D, 3, generator=g)
    m = "backward_map" if backward else "forward_map"
    pkeys = [f"warp.post_warp.{m}.linear_1.0.weight", f"warp.post_warp.{m}.linear_2.0.weight", f"warp.post_warp.{m}.linear_final.weight",
             f"warp.post_warp.{m}.linear_1.0.bias", f"warp.post_warp.{m}.linear_final.bias", "warp.skinning_model.delta_field.linear_1.0.weight"]

    def run(dev, mod):
        Pl = {k: (v.to(dev).clone().requires_grad_(True) if v.dtype.is_floating_point else v.to(dev)) for k, v in P.items()}
        frl = synthetic.to_device(fr, dev)
        dense = {k: v.clone().requires_grad_(True) for k, v in frl["dense"].items()}
        x = xyz.to(dev).clone().requires_grad_(True)
        te = frl["t_embed"] if backward else frl["t_embed_mean"]
        out, aux = mod.composed_warp(Pl, x, frl["t_articulation"], frl["rest_articulation"], te, frl["code_skin"], backward, dense=dense)
        loss = (out * w_out.to(dev)).sum()
        dk = ["t_embed", "code_bw" if backward else "code_fw"]
        gs = torch.autograd.grad(loss, [x] + [Pl[k] for k in pkeys] + [dense[k] for k in dk])
        return out, dict(zip(["x"] + pkeys + dk, gs))

    ro, rg = run("cpu", O)
    do, dg = run(DEV, warping)
    assert rel(do, ro) < 1e-4, f"xyz: {rel(do, ro):.3e}"
    for k in rg:
        assert rel(dg[k], rg[k]) < 1e-3, f"grad {k}: {rel(dg[k], rg[k]):.3e}"


def test_dense_warp_bf16_close_to_fp32():
    from lab4d_amd import mlp, warping
    M, N, D = 2, 64, 16
    P = synthetic.to_device(synthetic.add_dense_weights(synthetic.make_weights(4), seed=4), DEV)
    fr = synthetic.to_device(synthetic.add_codes(synthetic.make_frames(5, M, 64), synthetic.add_dense_weights(synthetic.make_weights(4), seed=4)), DEV)
    xyz = torch.randn(M, N, D, 3, device=DEV) * 0.08
    a = warping.dense_warp(P, xyz, fr["dense"]["t_embed"], fr["dense"]["code_bw"], True, mlp.PREC_F32)
    b = warping.dense_warp(P, xyz, fr["dense"]["t_embed"], fr["dense"]["code_bw"], True, mlp.PREC_BF16)
    assert float((a - b).abs().max()) < 2e-3  # motion is 0.1 * O(0.3): bf16 operand rounding


@pytest.mark.parametrize("prec_name", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 8, 64), (3, 5, 13)])
def test_fused_bone_coordinates_match_the_two_kernel_form(prec_name, shape, monkeypatch):
    """warping.SkinChain (bone coordinates formed inside the delta-skin chain kernel, lab4d_mlp_fwd_args.aff) against BoneCoords ->
    MlpChain on the same inputs: the (S,B) logits and every gradient.  (2,8,64): every 64-sample tile lies in one frame (affine rows
    staged in LDS); (3,5,13): tiles straddle frames (rows read per sample)."""
    from lab4d_amd import warping, mlp
    prec = mlp.PREC_F32 if prec_name == "f32" else mlp.PREC_BF16
    M, N, D = shape
    P0 = synthetic.make_weights(4)
    fr = synthetic.add_codes(synthetic.make_frames(5, M, 64), P0)
    g = torch.Generator().manual_seed(16)
    xyz = (torch.randn(M * N * D, 3, generator=g) * 0.08).to(DEV)
    w = torch.randn(M * N * D, 25, generator=g).to(DEV)
    pkeys = ["warp.skinning_model.log_gauss", "warp.skinning_model.delta_field.linear_1.0.weight", "warp.skinning_model.delta_field.linear_final.weight"]

    def run(fused):
        monkeypatch.setattr(warping, "FUSE_BONE_COORDS", fused)
        P = {k: (v.to(DEV).clone().requires_grad_(True) if v.dtype.is_floating_point else v.to(DEV)) for k, v in P0.items()}
        art = tuple(t.to(DEV).clone().requires_grad_(True) for t in fr["t_articulation"])
        x = xyz.clone().requires_grad_(True)
        raw, _ = warping.skin_logits(P, x, art, fr["t_embed"].to(DEV), fr["code_skin"].to(DEV), M, N * D, prec)
        gs = torch.autograd.grad((raw * w).sum(), [x, art[0], art[1]] + [P[k] for k in pkeys])
        return raw, gs

    r1, g1 = run(True)
    r0, g0 = run(False)
    tol = 1e-5 if prec_name == "f32" else 2e-2
    assert rel(r1, r0.cpu()) < tol
    for a, b, n in zip(g1, g0, ["x", "art_r", "art_d"] + pkeys):
        assert rel(a, b.cpu()) < tol, (n, rel(a, b.cpu()))
