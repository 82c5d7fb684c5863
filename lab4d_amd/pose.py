"""Host-side mirror of the per-frame pose / articulation path: lab4d/nnutils/pose.py
(`CameraMLP.get_vals`, `ArticulationSkelMLP.forward / get_vals_and_mean`), lab4d/nnutils/embedding.py (`TimeEmbedding`),
lab4d/nnutils/time.py (`TimeMLP`) and the forward kinematics of lab4d/utils/skel_utils.py.

What is native: the kinematic tree -- bone lengths, `so3_to_exp_map`, `fk_se3`, `matrix_to_quaternion`,
`shift_joints_to_bones_dq` and their adjoint -- is ONE gfx950 kernel each way (csrc/fk.hip, include/lab4d_pose.h) where
the reference runs a 25-step Python loop.  Round 6: the (M-row x W) MLPs in front of it -- TimeEmbedding (frame -> time
coordinate -> Fourier features -> mapping1 | InstEmbedding row -> mapping2), TimeMLP and the heads of CameraMLP /
IntrinsicsMLP / Articulation*MLP / AppearanceEmbedding -- are a PROGRAM of dense layers executed by csrc/rowmlp.hip
(include/lab4d_rowmlp.h, lab4d_amd/rowmlp.py): one launch forward, two backward per module, where the reference (and
rounds 1-5) issued one launch per nn.Linear / ReLU / cat / index.  Tensors on the GPU always take the kernels (no
fallback: a missing library raises); the torch algebra below is kept for CPU tensors only -- it is what the CPU suite
holds against the real reference modules (tests/test_patch_*.py) and what the kernels are held to on the GPU.  The
camera's quaternion product uses the library's quaternion_mul kernel.

Weights: flat dict keyed by the reference's state_dict names under a prefix (e.g. "warp.articulation" / "camera_mlp").
`info` holds TimeEmbedding's frame tables (embedding.py:153-175): frame_to_vid, frame_mapping, raw_fid_to_vid,
raw_fid_to_vidlen, raw_fid_to_vstart, max_ts, num_freq_t.  `skel` = {"rest_joints" (B,3), "edges" {child: parent, 1-based,
0 = root}, "symm_idx" [B]} = the buffers / attributes of ArticulationSkelMLP (pose.py:345-368).
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from . import rowmlp  # noqa: F401  (registers the lab4d_rowmlp_* signatures)
from .quat_utils import quaternion_mul

vp, ci = _lib.vp, _lib.ci
_lib.register("lab4d_fk_forward", [vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp])
_lib.register("lab4d_fk_backward", [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp])
_lib.register("lab4d_skel_bones_forward", [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp, vp, vp])
_lib.register("lab4d_skel_bones_backward", [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp])

i64_ = __import__("ctypes").c_int64
_lib.register("lab4d_camera_epilogue_forward", [vp, vp, vp, vp, ci, ci, vp, vp])
_lib.register("lab4d_camera_epilogue_backward", [vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, ci, vp])
_lib.register("lab4d_intrinsics_epilogue_forward", [vp, vp, vp, vp, vp, ci, ci, vp, vp])
_lib.register("lab4d_intrinsics_epilogue_backward", [vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp])

_SKEL_CACHE = {}


def skeleton_arrays(edges, B, device, symm_idx=None):
    """`edges` ({child: parent}, 1-based, 0 = root, iteration order = the reference's visiting order, skel_utils.py:82) ->
    int32 device arrays (order (B), parent (B), symm (B) or None)."""
    key = (tuple(edges.items()), B, str(device), None if symm_idx is None else tuple(int(s) for s in symm_idx))
    hit = _SKEL_CACHE.get(key)
    if hit is None:
        if len(edges) != B or sorted(edges.keys()) != list(range(1, B + 1)):
            raise RuntimeError("skeleton: edges must name every joint 1..%d exactly once" % B)
        order = torch.tensor([k - 1 for k in edges.keys()], dtype=torch.int32, device=device)
        parent = torch.tensor([edges[k] - 1 for k in range(1, B + 1)], dtype=torch.int32, device=device)
        symm = None if symm_idx is None else torch.tensor([int(s) for s in symm_idx], dtype=torch.int32, device=device)
        hit = _SKEL_CACHE[key] = (order, parent, symm)
    return hit


class _Fk(Function):
    """(so3 (R,B,3), local (R,B,3), shift (3)|None) -> dual quaternions ((R,B,4), (R,B,4)); bones: 0 = joints (fk_se3),
    1 = bone centres (fk_se3 + shift_joints_to_bones_dq)."""

    @staticmethod
    def forward(ctx, so3, local, shift, order, parent, bones):
        so3, local = so3.contiguous().float(), local.contiguous().float()
        shift_c = None if shift is None else shift.contiguous().float()
        _lib.require_device(so3, local, shift_c)
        R, B = so3.shape[:2]
        qr, qd = torch.empty(R, B, 4, device=so3.device), torch.empty(R, B, 4, device=so3.device)
        _lib.check(_lib.lib().lab4d_fk_forward(_lib.ptr(so3), _lib.ptr(local), _lib.ptr(shift_c), _lib.ptr(order), _lib.ptr(parent), R, B,
                                               int(bones), _lib.ptr(qr), _lib.ptr(qd), _lib.stream()), "fk_forward")
        ctx.save_for_backward(so3, local, shift_c if shift_c is not None else so3.new_empty(0), order, parent)
        ctx.meta = (int(bones), shift is not None)
        return qr, qd

    @staticmethod
    @once_differentiable
    def backward(ctx, g_qr, g_qd):
        so3, local, shift, order, parent = ctx.saved_tensors
        bones, has_shift = ctx.meta
        R, B = so3.shape[:2]
        g_qr, g_qd = g_qr.contiguous().float(), g_qd.contiguous().float()
        g_so3, g_local = torch.empty_like(so3), torch.empty_like(local)
        g_shift = torch.empty(R, 3, device=so3.device) if has_shift else None
        _lib.check(_lib.lib().lab4d_fk_backward(_lib.ptr(so3), _lib.ptr(local), _lib.ptr(shift) if has_shift else None, _lib.ptr(order),
                                                _lib.ptr(parent), _lib.ptr(g_qr), _lib.ptr(g_qd), R, B, bones, _lib.ptr(g_so3), _lib.ptr(g_local),
                                                _lib.ptr(g_shift), _lib.stream()), "fk_backward")
        return g_so3, g_local, (g_shift.sum(0) if has_shift else None), None, None, None


class _CameraEpilogue(Function):
    """CameraMLP.get_vals behind the heads (pose.py:126-147): quaternion_mul(F.normalize(raw), F.normalize(base_quat[video])) as ONE launch (two backward:
    the per-row adjoint, the per-video reduction of the base rotations' gradient -- added straight into a fused-accumulation sink when there is one)."""

    @staticmethod
    def forward(ctx, raw, base, frame_id, vid):
        raw, basec = raw.detach().contiguous().float(), base.detach().contiguous().float()
        _lib.require_device(raw, basec)
        M, V = raw.shape[0], basec.shape[0]
        out = torch.empty(M, 4, device=raw.device)
        _lib.check(_lib.lib().lab4d_camera_epilogue_forward(_lib.ptr(raw), _lib.ptr(basec), _lib.ptr(frame_id), _lib.ptr(vid), M, V, _lib.ptr(out), _lib.stream()),
                   "camera_epilogue_forward")
        ctx.save_for_backward(raw, basec, frame_id, vid)
        ctx.base_ref = base
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from . import mlp
        raw, base, frame_id, vid = ctx.saved_tensors
        M, V = raw.shape[0], base.shape[0]
        g = g.contiguous().float()
        g_raw, scratch = torch.empty_like(raw), torch.empty_like(raw)
        sink = mlp._grad_sink(ctx.base_ref) if ctx.needs_input_grad[1] else None
        g_base = sink if sink is not None else (torch.empty_like(base) if ctx.needs_input_grad[1] else None)
        _lib.check(_lib.lib().lab4d_camera_epilogue_backward(_lib.ptr(raw), _lib.ptr(base), _lib.ptr(frame_id), _lib.ptr(vid), _lib.ptr(g), M, V, _lib.ptr(g_raw),
                                                             _lib.ptr(scratch), _lib.ptr(g_base), int(sink is not None), _lib.stream()), "camera_epilogue_backward")
        return g_raw, (None if sink is not None else g_base), None, None


class _IntrinsicsEpilogue(Function):
    """IntrinsicsMLP.get_vals behind the head (intrinsics.py:94-107) as one launch each way (+ the per-video reduction)."""

    @staticmethod
    def forward(ctx, raw, logfocal, ppoint, frame_id, vid):
        raw, lf, pp = raw.detach().contiguous().float(), logfocal.detach().contiguous().float(), ppoint.detach().contiguous().float()
        _lib.require_device(raw, lf, pp)
        M, V = raw.shape[0], lf.shape[0]
        out = torch.empty(M, 4, device=raw.device)
        _lib.check(_lib.lib().lab4d_intrinsics_epilogue_forward(_lib.ptr(raw), _lib.ptr(lf), _lib.ptr(pp), _lib.ptr(frame_id), _lib.ptr(vid), M, V, _lib.ptr(out),
                                                                _lib.stream()), "intrinsics_epilogue_forward")
        ctx.save_for_backward(raw, lf, frame_id, vid)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        raw, lf, frame_id, vid = ctx.saved_tensors
        M, V = raw.shape[0], lf.shape[0]
        g = g.contiguous().float()
        g_raw, scratch, g_video = torch.empty_like(raw), torch.empty(M, 4, device=raw.device), torch.empty(V, 4, device=raw.device)
        _lib.check(_lib.lib().lab4d_intrinsics_epilogue_backward(_lib.ptr(raw), _lib.ptr(lf), _lib.ptr(frame_id), _lib.ptr(vid), _lib.ptr(g), M, V, _lib.ptr(g_raw),
                                                                 _lib.ptr(scratch), _lib.ptr(g_video), _lib.stream()), "intrinsics_epilogue_backward")
        return g_raw, g_video[:, :2], g_video[:, 2:], None, None


def _rows(x, B, C):
    return x.reshape(-1, B, C)


def fk_se3(local_rest_joints, so3, edges):
    """skel_utils.fk_se3(local_rest_joints, so3, edges, to_dq=True) (skel_utils.py:50-103): (..., B, 3) x2 ->
    ((..., B, 4), (..., B, 4)) joint-to-object dual quaternions."""
    B = so3.shape[-2]
    order, parent, _ = skeleton_arrays(edges, B, so3.device)
    qr, qd = _Fk.apply(_rows(so3, B, 3), _rows(local_rest_joints.expand_as(so3), B, 3), None, order, parent, 0)
    return qr.reshape(so3.shape[:-1] + (4,)), qd.reshape(so3.shape[:-1] + (4,))


def fk_bones(local_rest_joints, so3, edges, shift=None):
    """shift_joints_to_bones_dq(fk_se3(local_rest_joints, so3, edges), edges, shift) (skel_utils.py:50-145) in one launch."""
    B = so3.shape[-2]
    order, parent, _ = skeleton_arrays(edges, B, so3.device)
    qr, qd = _Fk.apply(_rows(so3, B, 3), _rows(local_rest_joints.expand_as(so3), B, 3), shift, order, parent, 1)
    return qr.reshape(so3.shape[:-1] + (4,)), qd.reshape(so3.shape[:-1] + (4,))


class _SkelBones(Function):
    """(so3 (R,B,3), loglen (R,B), logscale (1), rest_local (B,3), shift (3)) -> bone dual quaternions."""

    @staticmethod
    def forward(ctx, so3, loglen, logscale, rest_local, shift, order, parent, symm):
        so3, loglen = so3.contiguous().float(), loglen.contiguous().float()
        logscale, rest_local, shift = logscale.contiguous().float(), rest_local.contiguous().float(), shift.contiguous().float()
        _lib.require_device(so3, loglen, logscale, rest_local, shift)
        R, B = so3.shape[:2]
        qr, qd = torch.empty(R, B, 4, device=so3.device), torch.empty(R, B, 4, device=so3.device)
        _lib.check(_lib.lib().lab4d_skel_bones_forward(_lib.ptr(so3), _lib.ptr(loglen), _lib.ptr(logscale), _lib.ptr(rest_local), _lib.ptr(shift),
                                                       _lib.ptr(order), _lib.ptr(parent), _lib.ptr(symm), R, B, _lib.ptr(qr), _lib.ptr(qd),
                                                       _lib.stream()), "skel_bones_forward")
        ctx.save_for_backward(so3, loglen, logscale, rest_local, shift, order, parent, symm)
        return qr, qd

    @staticmethod
    @once_differentiable
    def backward(ctx, g_qr, g_qd):
        so3, loglen, logscale, rest_local, shift, order, parent, symm = ctx.saved_tensors
        R, B = so3.shape[:2]
        g_qr, g_qd = g_qr.contiguous().float(), g_qd.contiguous().float()
        g_so3, g_ll = torch.empty_like(so3), torch.empty_like(loglen)
        g_ls, g_sh = torch.empty(R, device=so3.device), torch.empty(R, 3, device=so3.device)
        _lib.check(_lib.lib().lab4d_skel_bones_backward(_lib.ptr(so3), _lib.ptr(loglen), _lib.ptr(logscale), _lib.ptr(rest_local), _lib.ptr(shift),
                                                        _lib.ptr(order), _lib.ptr(parent), _lib.ptr(symm), _lib.ptr(g_qr), _lib.ptr(g_qd), R, B,
                                                        _lib.ptr(g_so3), _lib.ptr(g_ll), _lib.ptr(g_ls), _lib.ptr(g_sh), _lib.stream()),
                   "skel_bones_backward")
        return g_so3, g_ll, g_ls.sum().reshape(logscale.shape), None, g_sh.sum(0), None, None, None


def rest_joints_to_local(rest_joints, edges):
    """skel_utils.rest_joints_to_local (skel_utils.py:35-47): child - parent for the edges whose parent is a joint."""
    pairs = [(c - 1, p - 1) for c, p in edges.items() if p > 0]
    idx = torch.tensor([c for c, _ in pairs], device=rest_joints.device)
    par = torch.tensor([p for _, p in pairs], device=rest_joints.device)
    local = rest_joints.clone()
    local[idx] = rest_joints[idx] - rest_joints[par]
    return local


def skel_bones(so3, log_bone_len_inc, logscale, skel, shift):
    """ArticulationSkelMLP.forward after the so3 head (pose.py:449-470): compute_rel_rest_joints (bone lengths
    exp(inc + logscale), symmetrised) + fk_se3 + shift_joints_to_bones_dq.  so3 (R,B,3); log_bone_len_inc (R,B) or (1,B)."""
    R, B = so3.shape[:2]
    order, parent, symm = skeleton_arrays(skel["edges"], B, so3.device, skel["symm_idx"])
    rest_local = rest_joints_to_local(skel["rest_joints"], skel["edges"])
    return _SkelBones.apply(so3, log_bone_len_inc.expand(R, B), logscale, rest_local, shift, order, parent, symm)


# ---- the per-frame modules around it ------------------------------------------------------------------------------------------


def _linear(P, name, x):
    return F.linear(x, P[name + ".weight"], P[name + ".bias"])


def _fourier(t, n_freq):
    """PosEmbedding(1, n_freq) (embedding.py:69-125) of a (M,1) time coordinate: [t, sin(2^k t), cos(2^k t)]_k."""
    if n_freq <= 0:
        return t
    ang = t * (2.0 ** torch.arange(n_freq, device=t.device, dtype=t.dtype))
    return torch.cat([t, torch.stack([torch.sin(ang), torch.cos(ang)], -1).reshape(t.shape[0], -1)], -1)


def frame_tid(frame_id, info):
    """embedding.py:177-184."""
    fid = frame_id.long()
    sub = frame_id - info["raw_fid_to_vstart"][fid]
    return (sub - info["raw_fid_to_vidlen"][fid] / 2) / info["max_ts"] * 2 * info.get("time_scale", 1.0)


def time_info_of(te):
    """`info` of a reference TimeEmbedding module (embedding.py:137-192): its frame tables, the normaliser max_ts its
    frame_to_tid closure captured (recovered from the tables: the longest video) and the number of Fourier bands."""
    dev = te.frame_mapping.device
    max_ts = float(te.raw_fid_to_vidlen.max())
    probe = te.frame_to_tid(torch.zeros(1, dtype=torch.long, device=dev))  # = (0 - len/2) / max_ts * 2 * time_scale
    time_scale = float(probe[0]) / (-(float(te.raw_fid_to_vidlen[0]) / 2) / max_ts * 2)
    return {"frame_to_vid": te.frame_to_vid, "frame_mapping": te.frame_mapping, "raw_fid_to_vid": te.raw_fid_to_vid,
            "raw_fid_to_vidlen": te.raw_fid_to_vidlen, "raw_fid_to_vstart": te.raw_fid_to_vstart, "max_ts": max_ts,
            "num_freq_t": te.fourier_embedding.N_freqs, "time_scale": time_scale}


def _vid_code(P, prefix, inst_id):
    w = P[prefix + ".inst_embedding.mapping.weight"]
    return w[torch.zeros_like(inst_id) if w.shape[0] == 1 else inst_id]


# ---- the same modules as rowmlp programs (GPU tensors) -------------------------------------------------------------------------


def _pad4(c):
    return (c + 3) // 4 * 4


def _lin_layer(P, name, src, dst, relu):
    return {"W": P[name + ".weight"], "b": P.get(name + ".bias"), "src": src, "dst": dst, "relu": relu}


def _time_prologue(P, te_prefix, frame_id, info):
    """The time prologue + TimeEmbedding's two Linear layers as program pieces: (time dict, layers, column of t_embed, width, next free column)."""
    fid = info["frame_mapping"] if frame_id is None else frame_id
    if not torch.is_tensor(fid):  # (the reference's frame_to_tid accepts lists / ints, embedding.py:178-179)
        fid = torch.as_tensor(fid)
    fid = fid.reshape(-1).long().to(info["raw_fid_to_vid"].device)  # (the reference indexes its device tables with whatever index tensor it is handed)
    info = dict(info, **{k: info[k].long() for k in ("raw_fid_to_vstart", "raw_fid_to_vidlen", "raw_fid_to_vid")})  # (no-ops on the reference's int64 buffers)
    nf = 2 * max(int(info["num_freq_t"]), 0) + 1
    w1, w2 = P[te_prefix + ".mapping1.weight"], P[te_prefix + ".mapping2.weight"]
    W = w1.shape[0]
    inst_W = P[te_prefix + ".inst_embedding.mapping.weight"]
    c_cat = _pad4(nf)
    c_te = c_cat + w2.shape[1]
    time = {"frame_id": fid, "vstart": info["raw_fid_to_vstart"], "vidlen": info["raw_fid_to_vidlen"], "vid": info["raw_fid_to_vid"],
            "max_ts": info["max_ts"], "time_scale": info.get("time_scale", 1.0), "n_freq": info["num_freq_t"], "four_col": 0, "inst_W": inst_W,
            "inst_col": c_cat + W}
    layers = [_lin_layer(P, te_prefix + ".mapping1", 0, c_cat, False), _lin_layer(P, te_prefix + ".mapping2", c_cat, c_te, False)]
    return time, layers, c_te, w2.shape[0], c_te + w2.shape[0], fid.shape[0]


def _time_mlp_layers(P, prefix, src, col, D):
    """TimeMLP.forward (time.py:65-73) as layers reading column `src`, writing from column `col`: (layers, column of the feature, width, next column)."""
    layers = []
    for i in range(D):
        name = f"{prefix}.linear_{i+1}.0"
        layers.append(_lin_layer(P, name, src, col, True))
        src, col = col, col + P[name + ".weight"].shape[0]
    name = f"{prefix}.linear_final.0"
    layers.append(_lin_layer(P, name, src, col, True))
    width = P[name + ".weight"].shape[0]
    return layers, col, width, col + width


def _head_layers(P, prefix, src, col):
    """nn.Sequential(Linear, ReLU, Linear) (pose.py:70-79): (layers, column of the output, width, next column)."""
    w0, w2 = P[prefix + ".0.weight"].shape[0], P[prefix + ".2.weight"].shape[0]
    return [_lin_layer(P, prefix + ".0", src, col, True), _lin_layer(P, prefix + ".2", col, col + w0, False)], col + w0, w2, _pad4(col + w0 + w2)


def _mlp_depth(P, prefix):
    D = 0
    while f"{prefix}.linear_{D+1}.0.weight" in P:
        D += 1
    return D


def _on_gpu(P, key):
    return P[key].is_cuda


def time_embedding(P, prefix, frame_id, info):
    """TimeEmbedding.forward (embedding.py:194-217)."""
    if _on_gpu(P, prefix + ".mapping1.weight"):
        time, layers, c_te, W, _, M = _time_prologue(P, prefix, frame_id, info)
        return rowmlp.run(layers, M, [(c_te, W)], time=time)[0]
    if frame_id is None:
        inst_id, t = info["frame_to_vid"], frame_tid(info["frame_mapping"], info)
    else:
        inst_id, t = info["raw_fid_to_vid"][frame_id], frame_tid(frame_id, info)
    coeff = _linear(P, prefix + ".mapping1", _fourier(t[:, None].float(), info["num_freq_t"]))
    return _linear(P, prefix + ".mapping2", torch.cat([coeff, _vid_code(P, prefix, inst_id)], -1))


def time_embedding_mean(P, prefix, info):
    """TimeEmbedding.get_mean_embedding (embedding.py:219-227)."""
    return time_embedding(P, prefix, info["frame_mapping"], info).mean(0, keepdim=True)


def time_mlp(P, prefix, t_embed, D=5):
    """TimeMLP.forward (time.py:65-73): D x (Linear + ReLU) + final Linear + ReLU."""
    if t_embed.is_cuda:
        W0 = t_embed.shape[-1]
        layers, c_f, Wf, _ = _time_mlp_layers(P, prefix, 0, _pad4(W0), D)
        return rowmlp.run(layers, t_embed.shape[0], [(c_f, Wf)], inputs=[((0, W0), t_embed)])[0]
    x = t_embed
    for i in range(D):
        x = F.relu(_linear(P, f"{prefix}.linear_{i+1}.0", x))
    return F.relu(_linear(P, f"{prefix}.linear_final.0", x))


def _head(P, prefix, x):
    return _linear(P, prefix + ".2", F.relu(_linear(P, prefix + ".0", x)))


def camera_vals(P, prefix, frame_id, info):
    """CameraMLP.get_vals (pose.py:116-147) -> (quat (M,4), trans (M,3))."""
    if _on_gpu(P, prefix + ".base_quat"):
        # ONE program: time prologue, TimeEmbedding, TimeMLP, both heads (12 layers)
        time, layers, c_te, _, col, M = _time_prologue(P, prefix + ".time_embedding", frame_id, info)
        l2, c_f, _, col = _time_mlp_layers(P, prefix, c_te, col, _mlp_depth(P, prefix))
        lt, c_t, w_t, col = _head_layers(P, prefix + ".trans", c_f, col)
        lq, c_q, w_q, col = _head_layers(P, prefix + ".quat", c_f, col)
        trans, quat = rowmlp.run(layers + l2 + lt + lq, M, [(c_t, w_t), (c_q, w_q)], time=time)
        return _CameraEpilogue.apply(quat, P[prefix + ".base_quat"], time["frame_id"], time["vid"]), trans
    feat = time_mlp(P, prefix, time_embedding(P, prefix + ".time_embedding", frame_id, info))
    quat = F.normalize(_head(P, prefix + ".quat", feat), dim=-1)
    inst_id = info["frame_to_vid"] if frame_id is None else info["raw_fid_to_vid"][frame_id]
    base = F.normalize(P[prefix + ".base_quat"][inst_id], dim=-1)
    return quaternion_mul(quat, base), _head(P, prefix + ".trans", feat)


def appearance_vals(P, prefix, frame_id, info):
    """AppearanceEmbedding.get_vals (time.py:107-117 + appearance.py:46-56): TimeEmbedding -> TimeMLP(D, W) -> Linear(W, C), one program on the GPU."""
    D = _mlp_depth(P, prefix)
    if _on_gpu(P, prefix + ".output.weight"):
        time, layers, c_te, _, col, M = _time_prologue(P, prefix + ".time_embedding", frame_id, info)
        l2, c_f, _, col = _time_mlp_layers(P, prefix, c_te, col, D)
        C = P[prefix + ".output.weight"].shape[0]
        return rowmlp.run(layers + l2 + [_lin_layer(P, prefix + ".output", c_f, col, False)], M, [(col, C)], time=time)[0]
    feat = time_mlp(P, prefix, time_embedding(P, prefix + ".time_embedding", frame_id, info), D=D)
    return _linear(P, prefix + ".output", feat)


def log_bone_len(P, prefix, inst_id, rows):
    """CondMLP(num_inst, in_channels=0, D=2, W=64) (pose.py:381-388): the instance code alone; inst_id None -> mean code."""
    w = P[prefix + ".inst_embedding.mapping.weight"]
    x = w.mean(0).expand(rows, -1) if inst_id is None else w[torch.zeros_like(inst_id) if w.shape[0] == 1 else inst_id]
    if x.is_cuda:  # one program: the code rows as an external input, 2 x (Linear + ReLU), Linear
        C = x.shape[-1]
        layers, src, col = [], 0, _pad4(C)
        for i in range(2):
            name = f"{prefix}.linear_{i+1}.0"
            layers.append(_lin_layer(P, name, src, col, True))
            src, col = col, col + P[name + ".weight"].shape[0]
        layers.append(_lin_layer(P, prefix + ".linear_final", src, col, False))
        return rowmlp.run(layers, x.shape[0], [(col, P[prefix + ".linear_final.weight"].shape[0])], inputs=[((0, C), x)])[0]
    for i in range(2):
        x = F.relu(_linear(P, f"{prefix}.linear_{i+1}.0", x))
    return _linear(P, prefix + ".linear_final", x)


def articulation_so3(P, prefix, t_embed):
    """pose.py:442-447: joint angles (M,B,3)."""
    if t_embed.is_cuda:
        W0 = t_embed.shape[-1]
        layers, c_f, _, col = _time_mlp_layers(P, prefix, 0, _pad4(W0), _mlp_depth(P, prefix))
        lh, c_o, w_o, _ = _head_layers(P, prefix + ".so3", c_f, col)
        return rowmlp.run(layers + lh, t_embed.shape[0], [(c_o, w_o)], inputs=[((0, W0), t_embed)])[0].reshape(t_embed.shape[0], -1, 3)
    return _head(P, prefix + ".so3", time_mlp(P, prefix, t_embed)).reshape(t_embed.shape[0], -1, 3)


def articulation_skel_forward(P, prefix, skel, t_embed, inst_id):
    """ArticulationSkelMLP.forward (pose.py:417-470) -> ((M,B,4), (M,B,4))."""
    so3 = articulation_so3(P, prefix, t_embed)
    ll = log_bone_len(P, prefix + ".log_bone_len", inst_id, so3.shape[0] if inst_id is not None else 1)
    return skel_bones(so3, ll, P[prefix + ".logscale"], skel, P[prefix + ".shift"])


def articulation_skel_vals_and_mean(P, prefix, skel, frame_id, info):
    """ArticulationSkelMLP.get_vals_and_mean (pose.py:526-573): frames and rest pose in one batched FK launch."""
    inst_id = info["frame_to_vid"] if frame_id is None else info["raw_fid_to_vid"][frame_id]
    bs = inst_id.shape[0]
    te = time_embedding(P, prefix + ".time_embedding", frame_id, info)
    te_mean = time_embedding_mean(P, prefix + ".time_embedding", info).expand(bs, -1)
    so3 = articulation_so3(P, prefix, torch.cat([te, te_mean], 0))
    ll = torch.cat([log_bone_len(P, prefix + ".log_bone_len", inst_id, bs), log_bone_len(P, prefix + ".log_bone_len", None, 1).expand(bs, -1)], 0)
    qr, qd = skel_bones(so3, ll, P[prefix + ".logscale"], skel, P[prefix + ".shift"])
    return (qr[:bs], qd[:bs]), (qr[bs:], qd[bs:])


def axis_angle_to_quaternion(aa):
    """quat_transform.py:149-174."""
    ang = aa.norm(dim=-1, keepdim=True)
    k = torch.where(ang.abs() < 1e-6, 0.5 - ang * ang / 48, torch.sin(ang * 0.5) / ang)
    return torch.cat([torch.cos(ang * 0.5), aa * k], -1)


def articulation_flat_forward(P, prefix, t_embed):
    """ArticulationFlatMLP.forward (pose.py:287-303), bag-of-bones motion ("bob"): ((M,B,4), (M,B,4)).  The dual part is
    0.5 * (0,t) x q on the library's quaternion_mul kernel (3-vector operand = pure quaternion, quaternion.cu:46-57)."""
    if t_embed.is_cuda:
        W0 = t_embed.shape[-1]
        layers, c_f, _, col = _time_mlp_layers(P, prefix, 0, _pad4(W0), _mlp_depth(P, prefix))
        lt, c_t, w_t, col = _head_layers(P, prefix + ".trans", c_f, col)
        ls, c_s, w_s, col = _head_layers(P, prefix + ".so3", c_f, col)
        tr, so3 = rowmlp.run(layers + lt + ls, t_embed.shape[0], [(c_t, w_t), (c_s, w_s)], inputs=[((0, W0), t_embed)])
    else:
        feat = time_mlp(P, prefix, t_embed)
        tr, so3 = _head(P, prefix + ".trans", feat), _head(P, prefix + ".so3", feat)
    trans = (tr * 0.1).reshape(t_embed.shape[0], -1, 3)
    qr = axis_angle_to_quaternion(so3.reshape(t_embed.shape[0], -1, 3))
    return qr, 0.5 * quaternion_mul(trans, qr)


def intrinsics_vals(P, prefix, frame_id, info):
    """IntrinsicsMLP.get_vals (intrinsics.py:86-107) -> (M,4) [fx, fy, px, py].  `info`: the module's own TimeEmbedding tables
    (num_freq_t = 0 and time_scale = 0.1 by default)."""
    if _on_gpu(P, prefix + ".base_logfocal"):
        time, layers, c_te, _, col, M = _time_prologue(P, prefix + ".time_embedding", frame_id, info)
        l2, c_f, _, col = _time_mlp_layers(P, prefix, c_te, col, _mlp_depth(P, prefix))
        lf, c_o, w_o, col = _head_layers(P, prefix + ".focal", c_f, col)
        raw = rowmlp.run(layers + l2 + lf, M, [(c_o, w_o)], time=time)[0]
        return _IntrinsicsEpilogue.apply(raw, P[prefix + ".base_logfocal"], P[prefix + ".base_ppoint"], time["frame_id"], time["vid"])
    else:
        focal = _head(P, prefix + ".focal", time_mlp(P, prefix, time_embedding(P, prefix + ".time_embedding", frame_id, info))).exp()
    inst_id = info["frame_to_vid"] if frame_id is None else info["raw_fid_to_vid"][frame_id]
    focal = focal * P[prefix + ".base_logfocal"][inst_id].exp()
    focal = (focal + focal.flip(-1)) / 2
    return torch.cat([focal, P[prefix + ".base_ppoint"][inst_id].expand_as(focal)], -1)
