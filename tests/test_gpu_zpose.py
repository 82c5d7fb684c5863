"""GPU parity of the per-frame articulation / camera path: the skeleton-FK kernels of csrc/fk.hip through
the C-ABI and the host layer lab4d_amd/pose.py, against the fixture the reference's own modules produced
(tests/golden/pose.pt) and against the oracle.  fp32, rtol 1e-4 (gradients 2e-4 of the tensor max)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def pose_fx(golden_dir):
    return torch.load(os.path.join(golden_dir, "pose.pt"), weights_only=False)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def dev_info(info):
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in info.items()}


def dev_skel(skel):
    return {"rest_joints": skel["rest_joints"].to(DEV), "edges": skel["edges"], "symm_idx": skel["symm_idx"]}


def test_fk_kernels_match_the_reference(pose_fx):
    from lab4d_amd import pose
    fk, edges = pose_fx["fk"], pose_fx["skel"]["edges"]
    so3, local, shift = (fk[k].to(DEV).requires_grad_(True) for k in ("so3", "local", "shift"))
    c = [x.to(DEV) for x in fk["cot"]]
    jr, jd = pose.fk_se3(local, so3, edges)
    assert rel(jr, fk["joints_dq"][0]) < 1e-5 and rel(jd, fk["joints_dq"][1]) < 1e-5
    g = torch.autograd.grad((jr * c[0]).sum() + (jd * c[1]).sum(), [so3, local])
    assert rel(g[0], fk["g_joints"][0]) < 2e-4 and rel(g[1], fk["g_joints"][1]) < 2e-4
    br, bd = pose.fk_bones(local, so3, edges, shift=shift)
    assert rel(br, fk["bones_dq"][0]) < 1e-5 and rel(bd, fk["bones_dq"][1]) < 1e-5
    g = torch.autograd.grad((br * c[2]).sum() + (bd * c[3]).sum(), [so3, local, shift])
    for a, b in zip(g, fk["g_bones"]):
        assert rel(a, b) < 2e-4


def test_articulation_skel_matches_the_reference(pose_fx):
    """ArticulationSkelMLP.get_vals_and_mean / get_vals: outputs and the gradient of every parameter."""
    from lab4d_amd import pose
    P = {"art." + k: (v.to(DEV).requires_grad_(True) if v.dtype.is_floating_point else v.to(DEV)) for k, v in pose_fx["art_state"].items()}
    skel, info, ref = dev_skel(pose_fx["skel"]), dev_info(pose_fx["time_info"]), pose_fx["art"]
    fid = pose_fx["frame_id"].to(DEV)
    assert rel(pose.time_embedding(P, "art.time_embedding", fid, info), ref["t_embed"]) < 1e-4
    assert rel(pose.articulation_so3(P, "art", ref["t_embed"].to(DEV)), ref["so3"]) < 1e-4
    (tr, td), (mr, md) = pose.articulation_skel_vals_and_mean(P, "art", skel, fid, info)
    for a, b in zip((tr, td, mr, md), ref["t"] + ref["mean"]):
        assert rel(a, b) < 1e-4
    cot = [x.to(DEV) for x in pose_fx["cot"]]
    ((tr * cot[0]).sum() + (td * cot[1]).sum() + (mr * cot[2]).sum() + (md * cot[3]).sum()).backward()
    for k, g in ref["grads"].items():
        assert rel(P["art." + k].grad, g) < 3e-4, k
    qr, qd = pose.articulation_skel_forward(P, "art", skel, pose.time_embedding(P, "art.time_embedding", None, info), info["frame_to_vid"])
    assert rel(qr, ref["all_frames"][0]) < 1e-4 and rel(qd, ref["all_frames"][1]) < 1e-4


def test_camera_matches_the_reference(pose_fx):
    from lab4d_amd import pose
    P = {"cam." + k: (v.to(DEV).requires_grad_(True) if v.dtype.is_floating_point else v.to(DEV)) for k, v in pose_fx["cam_state"].items()}
    info, ref = dev_info(pose_fx["time_info"]), pose_fx["cam"]
    q, t = pose.camera_vals(P, "cam", pose_fx["frame_id"].to(DEV), info)
    assert rel(q, ref["quat"]) < 1e-4 and rel(t, ref["trans"]) < 1e-4
    ((q * ref["cot"][0].to(DEV)).sum() + (t * ref["cot"][1].to(DEV)).sum()).backward()
    for k, g in ref["grads"].items():
        assert rel(P["cam." + k].grad, g) < 3e-4, k
    qa, ta = pose.camera_vals(P, "cam", None, info)
    assert rel(qa, ref["all_frames"][0]) < 1e-4 and rel(ta, ref["all_frames"][1]) < 1e-4


def test_fk_properties_at_scale(pose_fx):
    """Size-independent properties on 4,099 rows (ragged last block): unit real parts, zero angles give identity rotations
    with bone centres = the oracle's, and `shift` translates every bone centre by exactly `shift`."""
    from lab4d_amd import pose
    from oracle import pose_oracle as PO
    skel = pose_fx["skel"]
    edges, B, R = skel["edges"], skel["rest_joints"].shape[0], 4099
    g = torch.Generator().manual_seed(4)
    local = PO.rest_joints_to_local(skel["rest_joints"], edges)
    so3 = torch.randn(R, B, 3, generator=g) * 1.5
    so3[::7] = 0
    shift = torch.tensor([0.3, -0.2, 0.1])
    qr, qd = pose.fk_bones(local.to(DEV), so3.to(DEV), edges, shift=shift.to(DEV))
    qr0, qd0 = pose.fk_bones(local.to(DEV), so3.to(DEV), edges, shift=None)
    assert float((qr.norm(dim=-1) - 1).abs().max()) < 1e-5
    assert torch.equal(qr, qr0)
    t = 2 * PO.quaternion_mul(qd.cpu(), qr.cpu() * torch.tensor([1.0, -1, -1, -1]))[..., 1:]
    t0 = 2 * PO.quaternion_mul(qd0.cpu(), qr0.cpu() * torch.tensor([1.0, -1, -1, -1]))[..., 1:]
    assert float((t - t0 - shift).abs().max()) < 1e-5
    ident = qr[::7].cpu()
    assert float((ident - torch.tensor([1.0, 0, 0, 0])).abs().max()) < 1e-6
    ref_r, ref_d = PO.shift_joints_to_bones_dq(PO.fk_se3(local.expand(R, B, 3)[:64], so3[:64], edges), edges, shift=shift)
    assert rel(qr[:64], ref_r) < 1e-5 and rel(qd[:64], ref_d) < 1e-5


def test_flat_articulation_and_intrinsics_match_the_reference(pose_fx):
    from lab4d_amd import pose
    info, fid = dev_info(pose_fx["time_info"]), pose_fx["frame_id"].to(DEV)
    cot = [x.to(DEV) for x in pose_fx["cot"]]
    P = {"flat." + k: (v.to(DEV).requires_grad_(True) if v.dtype.is_floating_point else v.to(DEV)) for k, v in pose_fx["flat_state"].items()}
    qr, qd = pose.articulation_flat_forward(P, "flat", pose.time_embedding(P, "flat.time_embedding", fid, info))
    mr, md = pose.articulation_flat_forward(P, "flat", pose.time_embedding_mean(P, "flat.time_embedding", info))
    for a, b in zip((qr, qd, mr, md), pose_fx["flat"]["t"] + pose_fx["flat"]["mean"]):
        assert rel(a, b) < 1e-4
    ((qr * cot[0]).sum() + (qd * cot[1]).sum() + (mr * cot[2][:1]).sum() + (md * cot[3][:1]).sum()).backward()
    for k, g in pose_fx["flat"]["grads"].items():
        assert rel(P["flat." + k].grad, g) < 3e-4, k
    P = {"intr." + k: (v.to(DEV).requires_grad_(True) if v.dtype.is_floating_point else v.to(DEV)) for k, v in pose_fx["intr_state"].items()}
    info_k = dict(info, **pose_fx["intr_time"])
    kv = pose.intrinsics_vals(P, "intr", fid, info_k)
    assert rel(kv, pose_fx["intr"]["vals"]) < 1e-4
    (kv * pose_fx["intr"]["cot"].to(DEV)).sum().backward()
    for k, g in pose_fx["intr"]["grads"].items():
        assert rel(P["intr." + k].grad, g) < 3e-4, k
    assert rel(pose.intrinsics_vals(P, "intr", None, info_k), pose_fx["intr"]["all_frames"]) < 1e-4


def test_articulation_adapter_with_a_stand_in_module(pose_fx):
    """lab4d_amd.patch.articulation_skel_forward (what patch() binds to ArticulationSkelMLP.forward) driven by a stand-in module
    with the reference's parameter names: outputs incl. the return_so3 / override branches vs the reference-generated fixture."""
    from lab4d_amd import patch, pose
    from standins import Node, _tree
    ref, skel = pose_fx["art"], pose_fx["skel"]
    state = {k: v.to(DEV) for k, v in pose_fx["art_state"].items()}
    m = _tree(Node(), {k: v for k, v in state.items() if v.dtype.is_floating_point and k != "rest_joints"})
    m.register_buffer("rest_joints", skel["rest_joints"].to(DEV))
    m.edges, m.symm_idx, m.num_se3 = skel["edges"], skel["symm_idx"], skel["rest_joints"].shape[0]
    info = dev_info(pose_fx["time_info"])
    P = {"art." + k: v for k, v in state.items()}
    fid = pose_fx["frame_id"].to(DEV)
    te = pose.time_embedding(P, "art.time_embedding", fid, info)
    inst = info["raw_fid_to_vid"][fid]
    so3 = patch.articulation_skel_forward(m, te, inst, return_so3=True)
    assert rel(so3, ref["so3"]) < 1e-4
    qr, qd = patch.articulation_skel_forward(m, te, inst)
    assert rel(qr, ref["t"][0]) < 1e-4 and rel(qd, ref["t"][1]) < 1e-4
    # overrides (reanimation, pose.py:442-456): given joint angles, given bone-length increments, given local rest joints
    q2 = patch.articulation_skel_forward(m, te, inst, override_so3=ref["so3"].to(DEV))
    assert rel(q2[0], ref["t"][0]) < 1e-4 and rel(q2[1], ref["t"][1]) < 1e-4
    ll = pose.log_bone_len(P, "art.log_bone_len", inst, inst.shape[0])
    q3 = patch.articulation_skel_forward(m, te, inst, override_log_bone_len=ll)
    assert rel(q3[0], ref["t"][0]) < 1e-4 and rel(q3[1], ref["t"][1]) < 1e-4
    bl = (ll + P["art.logscale"]).exp()
    bl = (bl + bl[..., torch.as_tensor(skel["symm_idx"], device=DEV)]) / 2
    local = pose.rest_joints_to_local(m.rest_joints, skel["edges"])[None] * bl[..., None]
    q4 = patch.articulation_skel_forward(m, te, inst, override_local_rest_joints=local)
    assert rel(q4[0], ref["t"][0]) < 1e-4 and rel(q4[1], ref["t"][1]) < 1e-4
