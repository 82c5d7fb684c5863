"""GPU parity of csrc/rowmlp.hip (include/lab4d_rowmlp.h): programs of dense layers over per-row strips -- the per-frame MLPs of the
pose / appearance path as one launch forward and two backward.

* the program executor against the same layers in torch (fp32, 1e-5 relative): ragged widths, rows that do not fill a workgroup, fan-out (one
  feature, two heads), external inputs with gradients, the time prologue with one and with several videos;
* the reference's own modules: tests/test_gpu_zpose.py already holds pose.camera_vals / intrinsics_vals / articulation_* / time_embedding -- which run
  on these kernels for GPU tensors since round 6 -- to tests/golden/pose.pt (values and every parameter gradient); here the adapters patch() binds
  (TimeEmbedding.forward, CameraMLP.get_vals, IntrinsicsMLP.get_vals) are driven with stand-in modules against the same fixture;
* launch count: one kernel forward, two backward per module."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def leaf(*shape, g, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(DEV).requires_grad_(True)


@pytest.mark.parametrize("M,dims", [(1, (5, 7, 3)), (13, (13, 64, 64, 3)), (67, (199, 300, 64, 4)), (256, (256, 256, 256, 128, 7)), (9, (1024, 33, 1024, 2))])
def test_chain_matches_torch(M, dims):
    from lab4d_amd import rowmlp
    g = torch.Generator().manual_seed(sum(dims) + M)
    x = leaf(M, dims[0], g=g)
    Ws = [leaf(o, i, g=g, scale=i ** -0.5) for i, o in zip(dims[:-1], dims[1:])]
    bs = [leaf(o, g=g, scale=0.3) if k != 1 else None for k, o in enumerate(dims[1:])]  # (the second layer has no bias)
    col, layers, src = (dims[0] + 3) // 4 * 4, [], 0
    for k, (W, b) in enumerate(zip(Ws, bs)):
        layers.append({"W": W, "b": b, "src": src, "dst": col, "relu": k < len(Ws) - 1})
        src, col = col, col + W.shape[0]
    out = rowmlp.run(layers, M, [(src, dims[-1])], inputs=[((0, dims[0]), x)])[0]
    ref = x
    for k, (W, b) in enumerate(zip(Ws, bs)):
        ref = F.linear(ref, W, b)
        if k < len(Ws) - 1:
            ref = F.relu(ref)
    assert rel(out, ref) < 1e-5
    cot = torch.randn(M, dims[-1], generator=g).to(DEV)
    leaves = [x] + Ws + [b for b in bs if b is not None]
    ga = torch.autograd.grad((out * cot).sum(), leaves)
    gb = torch.autograd.grad((ref * cot).sum(), leaves)
    for a, b in zip(ga, gb):
        assert rel(a, b) < 2e-5


def test_fan_out_and_two_outputs():
    """One feature, two heads (CameraMLP's trans / quat, pose.py:70-79): the feature's gradient is the sum of both heads' input gradients."""
    from lab4d_amd import rowmlp
    g = torch.Generator().manual_seed(5)
    M, W = 37, 64
    x, W0, b0 = leaf(M, W, g=g), leaf(W, W, g=g, scale=0.2), leaf(W, g=g)
    Wa, ba, Wb, bb = leaf(3, W, g=g, scale=0.2), leaf(3, g=g), leaf(4, W, g=g, scale=0.2), leaf(4, g=g)
    layers = [{"W": W0, "b": b0, "src": 0, "dst": 64, "relu": True}, {"W": Wa, "b": ba, "src": 64, "dst": 128, "relu": False},
              {"W": Wb, "b": bb, "src": 64, "dst": 132, "relu": False}]
    a, b = rowmlp.run(layers, M, [(128, 3), (132, 4)], inputs=[((0, W), x)])
    f = F.relu(F.linear(x, W0, b0))
    ra, rb = F.linear(f, Wa, ba), F.linear(f, Wb, bb)
    assert rel(a, ra) < 1e-5 and rel(b, rb) < 1e-5
    ca, cb = torch.randn(M, 3, generator=g).to(DEV), torch.randn(M, 4, generator=g).to(DEV)
    leaves = [x, W0, b0, Wa, ba, Wb, bb]
    for u, v in zip(torch.autograd.grad((a * ca).sum() + (b * cb).sum(), leaves), torch.autograd.grad((ra * ca).sum() + (rb * cb).sum(), leaves, retain_graph=True)):
        assert rel(u, v) < 2e-5
    # only one of the two outputs used downstream: the other's gradient is None
    a2, _ = rowmlp.run(layers, M, [(128, 3), (132, 4)], inputs=[((0, W), x)])
    for u, v in zip(torch.autograd.grad((a2 * ca).sum(), [x, W0, Wa]), torch.autograd.grad((ra * ca).sum(), [x, W0, Wa])):
        assert rel(u, v) < 2e-5


@pytest.mark.parametrize("n_vid,n_freq,time_scale", [(1, 6, 1.0), (3, 4, 0.1), (2, 0, 1.0)])
def test_time_prologue_matches_the_torch_algebra(n_vid, n_freq, time_scale):
    """TimeEmbedding.forward (embedding.py:177-217) through the prologue against lab4d_amd.pose's torch algebra (held to the real reference on the CPU)."""
    from lab4d_amd import pose
    g = torch.Generator().manual_seed(n_vid * 10 + n_freq)
    W = 32
    lens = [40, 24, 17][:n_vid]
    off = [0]
    for n in lens:
        off.append(off[-1] + n)
    N = off[-1]
    vid = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(lens)])
    info = {"frame_to_vid": vid, "frame_mapping": torch.arange(N), "raw_fid_to_vid": vid, "raw_fid_to_vidlen": torch.tensor([lens[int(v)] for v in vid]),
            "raw_fid_to_vstart": torch.tensor([off[int(v)] for v in vid]), "max_ts": float(max(lens)), "num_freq_t": n_freq, "time_scale": time_scale}
    P0 = {"te.mapping1.weight": torch.randn(W, 2 * n_freq + 1, generator=g) * 0.5, "te.mapping1.bias": torch.randn(W, generator=g),
          "te.mapping2.weight": torch.randn(W, 2 * W, generator=g) * 0.2, "te.mapping2.bias": torch.randn(W, generator=g),
          "te.inst_embedding.mapping.weight": torch.randn(n_vid, W, generator=g)}
    fid = torch.tensor([0, N - 1, 5, 5, lens[0] - 1, N // 2, 3])
    cot = torch.randn(len(fid), W, generator=g)
    res = []
    for dev in ("cpu", DEV):
        P = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in P0.items()}
        inf = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in info.items()}
        te = pose.time_embedding(P, "te", fid.to(dev), inf)
        (te * cot.to(dev)).sum().backward()
        res.append((te, {k: v.grad for k, v in P.items()}, pose.time_embedding(P, "te", None, inf)))
    (a, ga, alla), (b, gb, allb) = res
    assert rel(b, a) < 2e-5 and rel(allb, alla) < 2e-5
    for k in ga:
        assert rel(gb[k], ga[k]) < 5e-5, k


@pytest.fixture(scope="module")
def pose_fx(golden_dir):
    return torch.load(os.path.join(golden_dir, "pose.pt"), weights_only=False)


def _time_module(state, prefix, fx_info):
    """Stand-in TimeEmbedding: parameters under the reference's names + the frame tables / closure the adapters read (embedding.py:137-192)."""
    from standins import Node, _tree
    te = _tree(Node(), {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix) and v.dtype.is_floating_point})
    for k in ("frame_to_vid", "frame_mapping", "raw_fid_to_vid", "raw_fid_to_vidlen", "raw_fid_to_vstart"):
        te.register_buffer(k, fx_info[k].to(DEV))
    max_ts, ts = float(fx_info["max_ts"]), float(fx_info.get("time_scale", 1.0))

    class Four:
        N_freqs = int(fx_info["num_freq_t"])
    te.fourier_embedding = Four()
    te.frame_to_tid = lambda f: ((f - te.raw_fid_to_vstart[f.long()]) - te.raw_fid_to_vidlen[f.long()] / 2) / max_ts * 2 * ts
    return te


def test_adapters_with_stand_in_modules(pose_fx):
    """What patch() binds to TimeEmbedding.forward / CameraMLP.get_vals / IntrinsicsMLP.get_vals, driven by stand-in modules with the reference's
    parameter names, against the reference-generated fixture (values; gradients through the module's own parameters)."""
    from lab4d_amd import patch
    from standins import Node, _tree
    fid = pose_fx["frame_id"].to(DEV)
    # camera
    state = {k: v.to(DEV) for k, v in pose_fx["cam_state"].items()}
    cam = _tree(Node(), {k: v for k, v in state.items() if v.dtype.is_floating_point and not k.startswith("time_embedding.")})
    cam.time_embedding = _time_module(state, "time_embedding.", pose_fx["time_info"])
    q, t = patch.camera_get_vals(cam, fid)
    ref = pose_fx["cam"]
    assert rel(q, ref["quat"]) < 1e-4 and rel(t, ref["trans"]) < 1e-4
    ((q * ref["cot"][0].to(DEV)).sum() + (t * ref["cot"][1].to(DEV)).sum()).backward()
    got = dict(cam.named_parameters())
    for k, gr in ref["grads"].items():
        assert rel(got[k].grad, gr) < 3e-4, k
    qa, ta = patch.camera_get_vals(cam)
    assert rel(qa, ref["all_frames"][0]) < 1e-4 and rel(ta, ref["all_frames"][1]) < 1e-4
    # the time embedding on its own, 1-D and (M, 1) frame ids
    te = patch.time_embedding_forward(cam.time_embedding, fid)
    te2 = patch.time_embedding_forward(cam.time_embedding, fid[:, None])
    assert te.shape == (len(fid), 64) and torch.equal(te, te2)
    # intrinsics
    state = {k: v.to(DEV) for k, v in pose_fx["intr_state"].items()}
    intr = _tree(Node(), {k: v for k, v in state.items() if v.dtype.is_floating_point and not k.startswith("time_embedding.")})
    intr.time_embedding = _time_module(state, "time_embedding.", dict(pose_fx["time_info"], **pose_fx["intr_time"]))
    kv = patch.intrinsics_get_vals(intr, fid)
    assert rel(kv, pose_fx["intr"]["vals"]) < 1e-4
    (kv * pose_fx["intr"]["cot"].to(DEV)).sum().backward()
    got = dict(intr.named_parameters())
    for k, gr in pose_fx["intr"]["grads"].items():
        assert rel(got[k].grad, gr) < 3e-4, k
    assert rel(patch.intrinsics_get_vals(intr), pose_fx["intr"]["all_frames"]) < 1e-4


def test_parameter_gradients_accumulate_into_fused_sinks():
    """With lab4d_amd.mlp.FUSED_GRAD_ACCUM (what patch.trainer_optimizer_init switches on) the parameter kernel ADDS dW / db / the InstEmbedding rows' gradient
    into `.grad` -- views of the optimizer's flat buffer -- and autograd is handed None: two backward passes sum, the values equal the returned-gradient
    path, parameters without a sink still get theirs from autograd."""
    from lab4d_amd import mlp, rowmlp
    g = torch.Generator().manual_seed(9)
    M = 19
    x = leaf(M, 12, g=g)
    W0, b0, W1 = leaf(20, 12, g=g), leaf(20, g=g), leaf(5, 20, g=g)
    layers = [{"W": W0, "b": b0, "src": 0, "dst": 12, "relu": True}, {"W": W1, "b": None, "src": 12, "dst": 32, "relu": False}]
    cot = torch.randn(M, 5, generator=g).to(DEV)
    ref = torch.autograd.grad((rowmlp.run(layers, M, [(32, 5)], inputs=[((0, 12), x)])[0] * cot).sum(), [x, W0, b0, W1])
    flat = torch.zeros(W0.numel() + b0.numel(), device=DEV)
    W0.grad, b0.grad = flat[:W0.numel()].view_as(W0), flat[W0.numel():].view_as(b0)  # sinks: contiguous fp32 views of one flat buffer; W1 has none
    old = mlp.FUSED_GRAD_ACCUM
    mlp.FUSED_GRAD_ACCUM = True
    try:
        for _ in range(2):
            (rowmlp.run(layers, M, [(32, 5)], inputs=[((0, 12), x)])[0] * cot).sum().backward()
    finally:
        mlp.FUSED_GRAD_ACCUM = old
    assert W0.grad.data_ptr() == flat.data_ptr(), "the sink was replaced instead of accumulated into"
    assert rel(W0.grad, 2 * ref[1]) < 1e-6 and rel(b0.grad, 2 * ref[2]) < 1e-6
    assert rel(W1.grad, 2 * ref[3]) < 1e-6 and rel(x.grad, 2 * ref[0]) < 1e-6


def test_one_launch_forward_two_backward(pose_fx):
    """The point of the row: launch count.  CameraMLP.get_vals' MLP part is ONE library launch forward and TWO backward (the torch modules:
    ~40 / ~100); counted with the library's own per-entry-point profile."""
    from lab4d_amd import _lib, pose
    P = {"cam." + k: (v.to(DEV).requires_grad_(True) if v.dtype.is_floating_point else v.to(DEV)) for k, v in pose_fx["cam_state"].items()}
    info = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in pose_fx["time_info"].items()}
    fid = pose_fx["frame_id"].to(DEV)
    pose.camera_vals(P, "cam", fid, info)  # warm-up
    _lib.PROF = {}
    try:
        q, t = pose.camera_vals(P, "cam", fid, info)
        (q.sum() + t.sum()).backward()
        torch.cuda.synchronize()
        prof = _lib.prof_summary()
    finally:
        _lib.PROF = None
    calls = {k: v[0] for k, v in prof.items()}
    assert calls.get("rowmlp_forward") == 1 and calls.get("rowmlp_backward") == 1, calls
    assert calls.get("camera_epilogue_forward") == 1 and calls.get("camera_epilogue_backward") == 1, calls
    assert "quaternion_mul_forward" not in calls  # (normalize x 2 + the product + their adjoints are the epilogue's two entry points now)
