"""CPU check of the HOST logic of the per-frame rowmlp programs (lab4d_amd/pose.py, round 6): the programs pose.py builds for TimeEmbedding /
CameraMLP / IntrinsicsMLP / Articulation*MLP / AppearanceEmbedding -- column plan, layer order, fan-out, prologue arguments -- are executed here by a
torch EMULATION of csrc/rowmlp.hip's contract (include/lab4d_rowmlp.h: a strip of columns per row, layers in index order, the time prologue) and must
reproduce pose.py's torch algebra, which tests/test_patch_*.py hold to the real reference modules.  No kernel runs here (the GPU suite holds the kernel to
the same algebra and to the reference's fixture: tests/test_gpu_rowmlp.py, tests/test_gpu_zpose.py); what this catches without a GPU is a wrong program."""
import os

import pytest
import torch
import torch.nn.functional as F

from lab4d_amd import pose, rowmlp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def emulate_run(layers, M, outs, time=None, inputs=(), row_stride=None):
    """The contract of lab4d_rowmlp_forward in torch (autograd does the backward)."""
    end = 0
    for L in layers:
        end = max(end, L["src"] + L["W"].shape[1], L["dst"] + L["W"].shape[0])
    for c, w in list(outs) + [cw for cw, _ in inputs]:
        end = max(end, c + w)
    written = []

    def claim(c, w, what):
        for c2, w2, what2 in written:
            assert c + w <= c2 or c2 + w2 <= c, "columns [%d,+%d) of %s overlap [%d,+%d) of %s" % (c, w, what, c2, w2, what2)
        written.append((c, w, what))

    strip = {}
    if time is not None:
        fid = time["frame_id"]
        assert fid.dtype == torch.int64 and fid.shape == (M,)
        sub = (fid - time["vstart"][fid]).float()
        t = (sub - time["vidlen"][fid].float() / 2.0) / float(time["max_ts"]) * 2.0 * float(time.get("time_scale", 1.0))
        nfq = max(int(time["n_freq"]), 0)
        cols = [t[:, None]]
        for k in range(nfq):
            cols += [torch.sin(2.0 ** k * t)[:, None], torch.cos(2.0 ** k * t)[:, None]]
        four = torch.cat(cols, -1)
        claim(time["four_col"], four.shape[1], "fourier")
        strip[(time["four_col"], four.shape[1])] = four
        iw = time.get("inst_W")
        if iw is not None:
            row = torch.zeros_like(fid) if iw.shape[0] == 1 else time["vid"][fid]
            claim(time["inst_col"], iw.shape[1], "inst")
            strip[(time["inst_col"], iw.shape[1])] = iw[row]
    for (c, w), x in inputs:
        claim(c, w, "input")
        strip[(c, w)] = x.as_subclass(torch.Tensor).reshape(M, w)  # (the tests mark CPU tensors as "on the GPU" with a subclass: drop the mark)

    def read(c, w):
        """columns [c, c + w) assembled from the pieces written so far (a concatenation = adjacent producers)"""
        parts, at = [], c
        while at < c + w:
            hit = [(k, v) for k, v in strip.items() if k[0] <= at < k[0] + k[1]]
            assert hit, "columns from %d are read before anything wrote them" % at
            (c0, w0), v = hit[0]
            take = min(c0 + w0, c + w) - at
            parts.append(v[:, at - c0: at - c0 + take])
            at += take
        return torch.cat(parts, -1)

    for i, L in enumerate(layers):
        W = L["W"]
        assert not (L["dst"] < L["src"] + W.shape[1] and L["src"] < L["dst"] + W.shape[0]), "layer %d writes into its input" % i
        y = F.linear(read(L["src"], W.shape[1]), W, L.get("b"))
        claim(L["dst"], W.shape[0], "layer %d" % i)
        strip[(L["dst"], W.shape[0])] = F.relu(y) if L.get("relu") else y
    return tuple(read(c, w) for c, w in outs)


class EmulatedCameraEpilogue:
    """The contract of lab4d_camera_epilogue_forward in torch."""

    @staticmethod
    def apply(raw, base, frame_id, vid):
        v = torch.zeros_like(frame_id) if base.shape[0] == 1 else vid[frame_id]
        from lab4d_amd.quat_utils import quaternion_mul
        return quaternion_mul(F.normalize(raw, dim=-1), F.normalize(base[v], dim=-1))


class EmulatedIntrinsicsEpilogue:
    @staticmethod
    def apply(raw, logfocal, ppoint, frame_id, vid):
        v = torch.zeros_like(frame_id) if logfocal.shape[0] == 1 else vid[frame_id]
        f = raw.exp() * logfocal[v].exp()
        f = (f + f.flip(-1)) / 2
        return torch.cat([f, ppoint[v].expand_as(f)], -1)


def emulate(monkeypatch):
    monkeypatch.setattr(rowmlp, "run", emulate_run)
    monkeypatch.setattr(pose, "_on_gpu", lambda P_, key: True)
    monkeypatch.setattr(pose, "_CameraEpilogue", EmulatedCameraEpilogue)
    monkeypatch.setattr(pose, "_IntrinsicsEpilogue", EmulatedIntrinsicsEpilogue)


@pytest.fixture(scope="module")
def fx():
    return torch.load(os.path.join(ROOT, "tests", "golden", "pose.pt"), weights_only=False)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def test_camera_intrinsics_time_programs_equal_the_algebra(fx, monkeypatch):
    info, fid = fx["time_info"], fx["frame_id"]
    P = {"cam." + k: v for k, v in fx["cam_state"].items()}
    ref = pose.camera_vals(P, "cam", fid, info)       # torch algebra (CPU tensors)
    ref_all = pose.camera_vals(P, "cam", None, info)
    te_ref = pose.time_embedding(P, "cam.time_embedding", fid, info)
    Pk = {"intr." + k: v for k, v in fx["intr_state"].items()}
    info_k = dict(info, **fx["intr_time"])
    k_ref = pose.intrinsics_vals(Pk, "intr", fid, info_k)
    emulate(monkeypatch)
    got = pose.camera_vals(P, "cam", fid, info)       # the rowmlp program, emulated
    got_all = pose.camera_vals(P, "cam", None, info)
    for a, b in zip(got + got_all, ref + ref_all):
        assert rel(a, b) < 1e-5
    assert rel(got[0], fx["cam"]["quat"]) < 1e-4 and rel(got[1], fx["cam"]["trans"]) < 1e-4  # ... and the reference's own values
    assert rel(pose.time_embedding(P, "cam.time_embedding", fid, info), te_ref) < 1e-5
    k_got = pose.intrinsics_vals(Pk, "intr", fid, info_k)
    assert rel(k_got, k_ref) < 1e-5 and rel(k_got, fx["intr"]["vals"]) < 1e-4


def test_articulation_and_appearance_programs_equal_the_algebra(fx, monkeypatch):
    info, fid = fx["time_info"], fx["frame_id"]
    P = {"art." + k: v for k, v in fx["art_state"].items()}
    te = pose.time_embedding(P, "art.time_embedding", fid, info)
    so3_ref = pose.articulation_so3(P, "art", te)
    Pf = {"flat." + k: v for k, v in fx["flat_state"].items()}
    tef = pose.time_embedding(Pf, "flat.time_embedding", fid, info)
    flat_ref = pose.articulation_flat_forward(Pf, "flat", tef)
    feat_ref = pose.time_mlp(P, "art", te)
    # an AppearanceEmbedding-shaped module: TimeMLP(D=2, W=64) + output Linear(64, 32), built from the camera's time embedding
    g = torch.Generator().manual_seed(0)
    Pa = {"appr.time_embedding." + k[len("time_embedding."):]: v for k, v in fx["cam_state"].items() if k.startswith("time_embedding.")}
    for i in (1, 2):
        Pa[f"appr.linear_{i}.0.weight"], Pa[f"appr.linear_{i}.0.bias"] = torch.randn(64, 64, generator=g) * 0.2, torch.randn(64, generator=g) * 0.1
    Pa["appr.linear_final.0.weight"], Pa["appr.linear_final.0.bias"] = torch.randn(64, 64, generator=g) * 0.2, torch.randn(64, generator=g) * 0.1
    Pa["appr.output.weight"], Pa["appr.output.bias"] = torch.randn(32, 64, generator=g) * 0.2, torch.randn(32, generator=g) * 0.1
    appr_ref = pose.appearance_vals(Pa, "appr", fid, info)

    emulate(monkeypatch)

    class Cuda(torch.Tensor):  # a CPU tensor that answers is_cuda = True: the tensor-input functions take their program branch
        @property
        def is_cuda(self):
            return True

    def cu(x):
        return x.as_subclass(Cuda)
    assert rel(pose.articulation_so3(P, "art", cu(te)), so3_ref) < 1e-5
    assert rel(pose.time_mlp(P, "art", cu(te)), feat_ref) < 1e-5
    got = pose.articulation_flat_forward(Pf, "flat", cu(tef))
    for a, b in zip(got, flat_ref):
        assert rel(a, b) < 1e-5
    assert rel(pose.appearance_vals(Pa, "appr", fid, info), appr_ref) < 1e-5
