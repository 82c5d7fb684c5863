"""GPU parity of the regularisation terms of dvr_model.compute_reg_loss against the fixtures the reference's
own methods produced (tests/golden/reg.pt, pose.pt): loss values and gradients, fp32."""
import os

import pytest
import torch

from lab4d_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def grad_ok(g, ref, tol=5e-4):
    g = g.detach().cpu()
    if "full" in ref:
        return rel(g, ref["full"]) < tol
    return rel(g.flatten()[::ref["stride"]], ref["sub"]) < tol and abs(float(g.double().norm()) - float(ref["norm"])) < tol * float(ref["norm"])


@pytest.fixture(scope="module")
def setup(golden_dir):
    fx = torch.load(os.path.join(golden_dir, "reg.pt"), weights_only=False)
    P = synthetic.to_device(synthetic.add_dense_weights(synthetic.make_weights(0, sdf_bias=fx["sdf_bias"])), DEV)
    for k, v in P.items():
        if v.dtype.is_floating_point and k != "aabb":
            v.requires_grad_(True)
    return fx, P


def test_visibility_decay(setup):
    from lab4d_amd import reg_losses as RL
    fx, P = setup
    v = fx["vis"]
    pts = RL.sample_points_aabb(fx["aabb"].to(DEV), v["u"].to(DEV), v["extend_factor"])
    table = P["vis_mlp.basefield.inst_embedding.mapping.weight"]
    for code in (table[v["inst_id"].to(DEV)], table[:1]):  # one code per point (n frames x 1 sample) and one shared code
        loss = RL.visibility_decay_loss(P, pts, code)
        assert rel(loss, v["loss"]) < 1e-4
        for (k, ref), g in zip(v["grads"].items(), torch.autograd.grad(loss, [P[k] for k in v["grads"]])):
            assert grad_ok(g, ref), k


def test_gauss_skin_consistency(setup):
    from lab4d_amd import reg_losses as RL
    fx, P = setup
    v = fx["gauss_skin"]
    art = tuple(x.to(DEV).requires_grad_(True) for x in v["rest_articulation_mean"])
    pts = RL.sample_points_aabb(fx["aabb"].to(DEV), v["u"].to(DEV), v["extend_factor"])
    loss = RL.gauss_skin_consistency_loss(P, pts, art, P["basefield.inst_embedding.mapping.weight"].mean(0, keepdim=True))
    assert rel(loss, v["loss"]) < 1e-4
    for a, b in zip(torch.autograd.grad(loss, list(art)), v["g_art"]):
        assert rel(a, b) < 5e-4


def test_soft_deform(setup):
    from lab4d_amd import reg_losses as RL
    fx, P = setup
    v = fx["soft_deform"]
    pts = RL.sample_points_aabb(fx["aabb"].to(DEV), v["u"].to(DEV), v["extend_factor"])
    inst = v["inst_id"].to(DEV)
    loss = RL.soft_deform_loss(P, pts, v["t_embed_dense"].to(DEV), P["warp.post_warp.forward_map.inst_embedding.mapping.weight"][inst],
                               P["warp.post_warp.backward_map.inst_embedding.mapping.weight"][inst])
    assert rel(loss, v["loss"]) < 2e-4
    for (k, ref), g in zip(v["grads"].items(), torch.autograd.grad(loss, [P[k] for k in v["grads"]])):
        assert grad_ok(g, ref, 1e-3), k


def test_priors(golden_dir):
    from lab4d_amd import reg_losses as RL
    fx = torch.load(os.path.join(golden_dir, "pose.pt"), weights_only=False)
    info = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in fx["time_info"].items()}
    P = {"art." + k: (v.to(DEV).requires_grad_(True) if v.dtype.is_floating_point else v.to(DEV)) for k, v in fx["art_state"].items()}
    loss = RL.skel_prior_loss(P, "art", info)
    assert rel(loss, fx["skel_prior"]["loss"]) < 1e-4
    loss.backward()
    for k, g in fx["skel_prior"]["grads"].items():
        got = P["art." + k].grad
        assert rel(got if got is not None else torch.zeros_like(g), g) < 5e-4 or float(g.abs().max()) == 0.0, k
    P = {"cam." + k: (v.to(DEV).requires_grad_(True) if v.dtype.is_floating_point else v.to(DEV)) for k, v in fx["cam_state"].items()}
    loss = RL.cam_prior_loss(P, "cam", info, fx["cam_prior"]["init_vals"].to(DEV))
    assert rel(loss, fx["cam_prior"]["loss"]) < 1e-4
    loss.backward()
    for k, g in fx["cam_prior"]["grads"].items():
        assert rel(P["cam." + k].grad, g) < 5e-4, k
