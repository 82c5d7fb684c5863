"""Golden vectors for the per-frame pose / articulation path, produced by the REFERENCE's own
modules on CPU: TimeEmbedding, CameraMLP.get_vals, ArticulationSkelMLP.{forward, get_vals_and_mean}, fk_se3,
shift_joints_to_bones_dq.

Run in the build container only (needs /root/reference):
    python tests/golden/make_pose_golden.py
Writes tests/golden/pose.pt.  The modules are built with W=64 to keep the fixture small (the algorithm does not depend
on the width); their state_dicts are stored in the fixture under the reference's own key names.
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
OUT_DIR = os.environ.get("LAB4D_GOLDEN_OUT", HERE)  # tests/test_golden_generator.py regenerates into a temp dir

from oracle import ref_shim  # noqa: E402

torch.set_num_threads(4)


def main():
    ref_shim.load()
    pose = importlib.import_module("lab4d.nnutils.pose")
    skel_utils = importlib.import_module("lab4d.utils.skel_utils")
    T = 64
    frame_info = {"frame_offset": np.asarray([0, 40, T]), "frame_offset_raw": np.asarray([0, 40, T]), "frame_mapping": list(range(T))}
    g = torch.Generator().manual_seed(11)
    out = {"frame_info": {k: (v.tolist() if hasattr(v, "tolist") else list(v)) for k, v in frame_info.items()}}

    # ---- skeleton articulation ------------------------------------------------------------------------------------
    torch.manual_seed(3)
    art = pose.ArticulationSkelMLP(frame_info, "quad", None, W=64)
    with torch.no_grad():
        art.so3[2].weight.mul_(6.0)  # joint angles of O(1) rad instead of the small default init
        art.shift.copy_(torch.tensor([0.01, -0.02, 0.03]))
        art.logscale.fill_(-0.3)
        art.log_bone_len.linear_final.weight.mul_(3.0)
    out["skel"] = {"rest_joints": art.rest_joints.clone(), "edges": dict(art.edges), "symm_idx": list(art.symm_idx)}
    out["art_state"] = {k: v.clone() for k, v in art.state_dict().items()}
    te = art.time_embedding
    out["time_info"] = {"num_freq_t": int(te.fourier_embedding.num_freqs) if hasattr(te.fourier_embedding, "num_freqs")
                        else (te.fourier_embedding.out_channels - 1) // 2,
                        "frame_to_vid": te.frame_to_vid.clone(), "frame_mapping": te.frame_mapping.clone(),
                        "raw_fid_to_vid": te.raw_fid_to_vid.clone(), "raw_fid_to_vidlen": te.raw_fid_to_vidlen.clone(),
                        "raw_fid_to_vstart": te.raw_fid_to_vstart.clone(),
                        "max_ts": float((frame_info["frame_offset_raw"][1:] - frame_info["frame_offset_raw"][:-1]).max())}
    fid = torch.tensor([3, 4, 39, 40, 63, 17])
    out["frame_id"] = fid
    cot = [torch.randn(len(fid), 25, 4, generator=g) for _ in range(4)]
    out["cot"] = cot
    art.zero_grad()
    t_embed = te(fid)
    (tr, td), (mr, md) = art.get_vals_and_mean(fid)
    loss = (tr * cot[0]).sum() + (td * cot[1]).sum() + (mr * cot[2]).sum() + (md * cot[3]).sum()
    loss.backward()
    out["art"] = {"t_embed": t_embed.detach().clone(), "t_embed_mean": te.get_mean_embedding("cpu").detach().clone(),
                  "so3": art.forward(t_embed, te.raw_fid_to_vid[fid], return_so3=True).detach().clone(),
                  "rel_rest_joints_inst": art.compute_rel_rest_joints(inst_id=te.raw_fid_to_vid[fid]).detach().clone(),
                  "rel_rest_joints_mean": art.compute_rel_rest_joints().detach().clone(),
                  "t": (tr.detach().clone(), td.detach().clone()), "mean": (mr.detach().clone(), md.detach().clone()),
                  "all_frames": tuple(x.detach().clone() for x in art.get_vals()),
                  "grads": {k: p.grad.clone() for k, p in art.named_parameters() if p.grad is not None}}

    # ---- forward kinematics at op level: large angles (every matrix_to_quaternion branch), zero angles (theta clamp) ----
    R, B = 48, 25
    so3 = torch.randn(R, B, 3, generator=g) * 1.6
    so3[0] = 0
    so3[1, ::2] = 0
    so3[2] = so3[2] / so3[2].norm(dim=-1, keepdim=True) * 3.1  # close to pi: the w-branch is not the best conditioned one
    local = (skel_utils.rest_joints_to_local(art.rest_joints, art.edges)[None] * (0.5 + torch.rand(R, B, 1, generator=g))).contiguous()
    shift = torch.tensor([0.05, 0.02, -0.01])
    so3.requires_grad_(True), local.requires_grad_(True), shift.requires_grad_(True)
    jr, jd = skel_utils.fk_se3(local, so3, art.edges)
    br, bd = skel_utils.shift_joints_to_bones_dq((jr, jd), art.edges, shift=shift)
    c = [torch.randn(R, B, 4, generator=g) for _ in range(4)]
    gj = torch.autograd.grad((jr * c[0]).sum() + (jd * c[1]).sum(), [so3, local], retain_graph=True)
    gb = torch.autograd.grad((br * c[2]).sum() + (bd * c[3]).sum(), [so3, local, shift])
    out["fk"] = {"so3": so3.detach().clone(), "local": local.detach().clone(), "shift": shift.detach().clone(), "cot": c,
                 "joints_dq": (jr.detach().clone(), jd.detach().clone()), "bones_dq": (br.detach().clone(), bd.detach().clone()),
                 "g_joints": tuple(x.clone() for x in gj), "g_bones": tuple(x.clone() for x in gb)}
    # which candidate matrix_to_quaternion picked: the fixture must cover all four
    G = skel_utils.fk_se3(local.detach(), so3.detach(), art.edges, to_dq=False)[..., :3, :3]
    d = torch.stack([1 + G[..., 0, 0] + G[..., 1, 1] + G[..., 2, 2], 1 + G[..., 0, 0] - G[..., 1, 1] - G[..., 2, 2],
                     1 - G[..., 0, 0] + G[..., 1, 1] - G[..., 2, 2], 1 - G[..., 0, 0] - G[..., 1, 1] + G[..., 2, 2]], -1)
    out["fk"]["branch_hist"] = torch.bincount(d.argmax(-1).flatten(), minlength=4)
    assert (out["fk"]["branch_hist"] > 0).all(), out["fk"]["branch_hist"]

    # ---- camera --------------------------------------------------------------------------------------------------
    torch.manual_seed(5)
    rtmat = ref_shim.synthetic_data_info(T)["rtmat"]
    cam = pose.CameraMLP(rtmat, frame_info=frame_info, W=64)
    with torch.no_grad():
        cam.base_quat.copy_(torch.randn(2, 4, generator=g))
    out["cam_state"] = {k: v.clone() for k, v in cam.state_dict().items()}
    cam.zero_grad()
    q, t = cam.get_vals(fid)
    cq, ct = torch.randn(len(fid), 4, generator=g), torch.randn(len(fid), 3, generator=g)
    ((q * cq).sum() + (t * ct).sum()).backward()
    out["cam"] = {"quat": q.detach().clone(), "trans": t.detach().clone(), "cot": (cq, ct),
                  "all_frames": tuple(x.detach().clone() for x in cam.get_vals()),
                  "grads": {k: p.grad.clone() for k, p in cam.named_parameters() if p.grad is not None}}
    # ---- the two priors of compute_reg_loss that live on these modules (engine/model.py:525-526) ----
    art.zero_grad()
    loss = art.skel_prior_loss()
    loss.backward()
    out["skel_prior"] = {"loss": loss.detach().clone(), "grads": {k: p.grad.clone() for k, p in art.named_parameters() if p.grad is not None}}
    cam.zero_grad()
    loss = cam.compute_distance_to_prior()
    loss.backward()
    out["cam_prior"] = {"loss": loss.detach().clone(), "init_vals": cam.init_vals.clone(),
                        "grads": {k: p.grad.clone() for k, p in cam.named_parameters() if p.grad is not None}}
    # ---- bag-of-bones articulation (fg_motion "bob") and intrinsics ----
    intrinsics_mod = importlib.import_module("lab4d.nnutils.intrinsics")
    torch.manual_seed(7)
    flat = pose.ArticulationFlatMLP(frame_info, 25, W=64)
    with torch.no_grad():
        flat.so3[2].weight.mul_(6.0)
    out["flat_state"] = {k: v.clone() for k, v in flat.state_dict().items()}
    flat.zero_grad()
    fr_, fd_ = flat.get_vals(fid)
    mr_, md_ = flat.get_mean_vals()
    ((fr_ * cot[0]).sum() + (fd_ * cot[1]).sum() + (mr_ * cot[2][:1]).sum() + (md_ * cot[3][:1]).sum()).backward()
    out["flat"] = {"t": (fr_.detach().clone(), fd_.detach().clone()), "mean": (mr_.detach().clone(), md_.detach().clone()),
                   "grads": {k: p.grad.clone() for k, p in flat.named_parameters() if p.grad is not None}}
    torch.manual_seed(9)
    intr = intrinsics_mod.IntrinsicsMLP(np.tile(np.asarray([[64.0, 64.0, 32.0, 32.0]], dtype=np.float32), (T, 1)), frame_info=frame_info, W=64)
    with torch.no_grad():
        intr.base_logfocal.copy_(torch.tensor([[4.1, 4.2], [4.0, 3.9]]))
        intr.base_ppoint.copy_(torch.tensor([[32.0, 31.0], [30.0, 33.0]]))
    ti = intr.time_embedding
    out["intr_state"] = {k: v.clone() for k, v in intr.state_dict().items()}
    out["intr_time"] = {"num_freq_t": (ti.fourier_embedding.out_channels - 1) // 2, "time_scale": 0.1}
    intr.zero_grad()
    kv = intr.get_vals(fid)
    ck = torch.randn(len(fid), 4, generator=g)
    (kv * ck).sum().backward()
    out["intr"] = {"vals": kv.detach().clone(), "cot": ck, "all_frames": intr.get_vals().detach().clone(),
                   "grads": {k: p.grad.clone() for k, p in intr.named_parameters() if p.grad is not None}}
    # ---- appearance code: AppearanceEmbedding.get_vals = Fourier(t) -> TimeEmbedding -> TimeMLP(D=2, W=64) -> Linear(64, 32) ----
    appearance = importlib.import_module("lab4d.nnutils.appearance")
    torch.manual_seed(13)
    ae = appearance.AppearanceEmbedding(frame_info, 32)
    with torch.no_grad():
        for p_ in ae.parameters():  # away from the init (zero biases) so that every term counts
            p_.add_(0.05 * torch.randn_like(p_))
    ta = ae.time_embedding
    probe = ta.frame_to_tid(torch.zeros(1, dtype=torch.long))
    max_ts = float(ta.raw_fid_to_vidlen.max())
    out["appr_state"] = {k: v.clone() for k, v in ae.state_dict().items()}
    out["appr_time"] = {"num_freq_t": int(ta.fourier_embedding.N_freqs), "max_ts": max_ts,
                        "time_scale": float(probe[0]) / (-(float(ta.raw_fid_to_vidlen[0]) / 2) / max_ts * 2)}
    ae.zero_grad()
    av = ae.get_vals(fid)
    ca = torch.randn(len(fid), 32, generator=g)
    (av * ca).sum().backward()
    out["appr"] = {"vals": av.detach().clone(), "cot": ca, "all_frames": ae.get_vals().detach().clone(),
                   "grads": {k: p_.grad.clone() for k, p_ in ae.named_parameters() if p_.grad is not None}}
    path = os.path.join(OUT_DIR, "pose.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; matrix_to_quaternion branches:", out["fk"]["branch_hist"].tolist())


if __name__ == "__main__":
    main()
