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
@pytest.mark.parametrize("n_bones", [25, 18])
def test_delta_skin_chain_forms_match_the_two_kernel_form(prec_name, shape, n_bones, monkeypatch):
    """The three forms of the delta-skin field on the same inputs -- the (S,B) logits and every gradient, against BoneCoords -> MlpChain:
    "fused": warping.SkinChain (bone coordinates formed inside the chain kernel, lab4d_mlp_fwd_args.aff);
    "affine": warping.SkinChainA (linear_1 folded into a per-frame table of the point, LAB4D_NET_SKIN_A / _SKIN18_A: the adjoint of that
    layer -- point gradient and per-frame table gradient -- is taken inside the backward chain kernel).
    (2,8,64): every 64-sample tile lies in one frame (rows staged in LDS, table gradient reduced in registers); (3,5,13): tiles straddle
    frames (rows read per sample, element-wise atomics)."""
    from lab4d_amd import warping, mlp
    prec = mlp.PREC_F32 if prec_name == "f32" else mlp.PREC_BF16
    M, N, D = shape
    P0 = synthetic.make_weights(4, num_bones=n_bones)
    fr = synthetic.add_codes(synthetic.make_frames(5, M, 64, num_bones=n_bones), P0)
    g = torch.Generator().manual_seed(16)
    xyz = (torch.randn(M * N * D, 3, generator=g) * 0.08).to(DEV)
    w = torch.randn(M * N * D, n_bones, generator=g).to(DEV)
    q = "warp.skinning_model.delta_field."
    pkeys = ["warp.skinning_model.log_gauss", q + "linear_1.0.weight", q + "linear_1.0.bias", q + "linear_2.0.weight", q + "linear_2.0.bias",
             q + "linear_final.weight", q + "linear_final.bias"]

    def run(form):
        monkeypatch.setattr(warping, "FUSE_BONE_COORDS", form != "two_kernel")
        monkeypatch.setattr(warping, "SKIN_AFFINE", form == "affine")
        P = {k: (v.to(DEV).clone().requires_grad_(True) if v.dtype.is_floating_point else v.to(DEV)) for k, v in P0.items()}
        art = tuple(t.to(DEV).clone().requires_grad_(True) for t in fr["t_articulation"])
        te, code = fr["t_embed"].to(DEV).clone().requires_grad_(True), fr["code_skin"].to(DEV).clone().requires_grad_(True)
        x = xyz.clone().requires_grad_(True)
        raw, _ = warping.skin_logits(P, x, art, te, code, M, N * D, prec)
        gs = torch.autograd.grad((raw * w).sum(), [x, art[0], art[1], te, code] + [P[k] for k in pkeys])
        return raw, gs

    names = ["x", "art_r", "art_d", "t_embed", "code"] + pkeys
    if prec_name == "f32":
        r0, g0 = run("two_kernel")
        for form in ("fused", "affine"):
            r1, g1 = run(form)
            assert rel(r1, r0.cpu()) < 1e-5, (form, rel(r1, r0.cpu()))
            for a, b, n in zip(g1, g0, names):
                assert rel(a, b.cpu()) < 1e-5, (form, n, rel(a, b.cpu()))
        return
    # bf16: the affine form evaluates linear_1 in fp32 while the other two round the coordinates and linear_1 to bf16, so units whose
    # pre-activation is within a bf16 ulp of zero switch differently between the forms (an O(1) difference in single samples' gradients):
    # every form is held to the fp32 result in the L2 norm instead
    def rel2(a, b):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        return float((a - b).norm() / (b.norm() + 1e-12))

    prec = mlp.PREC_F32  # run() reads it at call time
    r0, g0 = run("two_kernel")
    prec = mlp.PREC_BF16
    err = {}
    for form in ("two_kernel", "fused", "affine"):
        r1, g1 = run(form)
        err[form] = {"raw": rel2(r1, r0), **{n: rel2(a, b) for a, b, n in zip(g1, g0, names)}}
    for n in ["raw"] + names:
        base = max(err["two_kernel"][n], err["fused"][n])
        assert base < 1e-1, (n, err)  # the bf16 path itself (units within a bf16 ulp of zero switch against fp32)
        assert err["affine"][n] < max(2e-2, 1.5 * base), (n, err)  # the affine form is no further from fp32 than the coordinate forms


def test_delta_skin_affine_form_inference_matches_training_forward():
    """no_grad (the inference-mode kernel, nothing stored) and the training-mode forward of the affine form agree bit for bit."""
    from lab4d_amd import warping, mlp
    M, N, D = 2, 8, 64
    P0 = synthetic.make_weights(4)
    fr = synthetic.add_codes(synthetic.make_frames(5, M, 64), P0)
    P = {k: (v.to(DEV).clone().requires_grad_(True) if v.dtype.is_floating_point else v.to(DEV)) for k, v in P0.items()}
    art = tuple(t.to(DEV) for t in fr["t_articulation"])
    x = (torch.randn(M * N * D, 3) * 0.08).to(DEV)
    for prec in (mlp.PREC_F32, mlp.PREC_BF16):
        a, _ = warping.skin_logits(P, x, art, fr["t_embed"].to(DEV), fr["code_skin"].to(DEV), M, N * D, prec)
        with torch.no_grad():
            b, _ = warping.skin_logits(P, x, art, fr["t_embed"].to(DEV), fr["code_skin"].to(DEV), M, N * D, prec)
        assert torch.equal(a.detach(), b)


@pytest.mark.parametrize("shape", [(2, 4, 64), (3, 5, 13)])
def test_forward_multi_equals_separate_forward_warps(shape):
    """skinning_warp_forward_multi (one delta-skin evaluation, the blends' adjoints accumulated into one gradient pair by
    lab4d_skin_blend_backward_acc) against two independent skinning_warp(backward=False) calls: outputs bit for bit, gradients to rounding."""
    from lab4d_amd import warping, mlp
    M, N, D = shape
    P0 = synthetic.make_weights(4)
    fr = synthetic.add_codes(synthetic.make_frames(5, M, 64), P0)
    g = torch.Generator().manual_seed(3)
    xyz = torch.randn(M, N, D, 3, generator=g) * 0.08
    ws = [torch.randn(M, N, D, 3, generator=g).to(DEV) for _ in range(2)]
    we = [torch.randn(M, N, D, 1, generator=g).to(DEV) for _ in range(2)]
    pkeys = ["warp.skinning_model.log_gauss", "warp.skinning_model.delta_field.linear_1.0.weight", "warp.skinning_model.delta_field.linear_final.weight"]

    def run(multi):
        P = {k: (v.to(DEV).clone().requires_grad_(True) if v.dtype.is_floating_point else v.to(DEV)) for k, v in P0.items()}
        t_art = tuple(t.to(DEV).clone().requires_grad_(True) for t in fr["t_articulation"])
        nxt = tuple(t.to(DEV).flip(0).clone().requires_grad_(True) for t in fr["t_articulation"])
        rest = tuple(t.to(DEV).clone().requires_grad_(True) for t in fr["rest_articulation"])
        x = xyz.to(DEV).clone().requires_grad_(True)
        te, code = fr["t_embed_mean"].to(DEV), fr["code_skin"].to(DEV)
        if multi:
            res = warping.skinning_warp_forward_multi(P, x, [nxt, t_art], rest, te, code, mlp.PREC_F32)
        else:
            res = [warping.skinning_warp(P, x, a, rest, te, code, False, mlp.PREC_F32) for a in (nxt, t_art)]
        loss = sum((o * w).sum() + (aux["skin_entropy"] * e).sum() + (aux["delta_skin"] * e).sum() * 50 for (o, aux), w, e in zip(res, ws, we))
        gs = torch.autograd.grad(loss, [x, t_art[0], t_art[1], nxt[0], nxt[1], rest[0], rest[1]] + [P[k] for k in pkeys])
        return [o for o, _ in res] + [aux[k] for _, aux in res for k in ("skin_entropy", "delta_skin")], gs

    o1, g1 = run(True)
    o0, g0 = run(False)
    for a, b in zip(o1, o0):
        assert torch.equal(a, b)
    for a, b, n in zip(g1, g0, ["x", "t_r", "t_d", "n_r", "n_d", "rest_r", "rest_d"] + pkeys):
        assert rel(a, b.cpu()) < 1e-5, (n, rel(a, b.cpu()))
