"""Host side of csrc/rowmlp.hip (include/lab4d_rowmlp.h): the per-frame (M-row) MLPs in front of the hot path -- TimeEmbedding, TimeMLP and the
heads of CameraMLP / IntrinsicsMLP / ArticulationSkelMLP / ArticulationFlatMLP / AppearanceEmbedding (lab4d/nnutils/embedding.py:
177-217, base.py:65-78, time.py:65-73, pose.py:103-147,442-447, intrinsics.py:73-107, appearance.py:46-56) -- as ONE launch forward and TWO backward
per module instead of one torch launch per nn.Linear / ReLU / cat / index (~40 forward, ~100 backward).

`run(layers, M, outs, time=..., inputs=...)` executes a program of dense layers over a per-row strip of floats (see the header for the data model)
and returns the requested column ranges as autograd-connected tensors: gradients flow to every weight / bias, to the InstEmbedding table of the time
prologue and to the `inputs`.  No fallback: CUDA tensors only (`_lib.require_device`), the library must be built.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib

MAX_LAYERS = 16
vp, ci, cf, i32 = _lib.vp, _lib.ci, _lib.cf, ctypes.c_int32


class _Layer(ctypes.Structure):
    _fields_ = [("W", vp), ("b", vp), ("dW", vp), ("db", vp), ("in_dim", i32), ("out_dim", i32), ("src_col", i32), ("dst_col", i32), ("relu", i32),
                ("pad_", i32)]


class _Prog(ctypes.Structure):
    _fields_ = [("n_layers", i32), ("row_stride", i32), ("frame_id", vp), ("vstart", vp), ("vidlen", vp), ("vid", vp), ("inst_W", vp), ("d_inst_W", vp),
                ("max_ts", cf), ("time_scale", cf), ("n_freq", i32), ("four_col", i32), ("inst_col", i32), ("inst_dim", i32), ("inst_rows", i32),
                ("pad_", i32), ("layer", _Layer * MAX_LAYERS)]


_lib.register("lab4d_rowmlp_forward", [ctypes.POINTER(_Prog), vp, ci, vp])
_lib.register("lab4d_rowmlp_backward", [ctypes.POINTER(_Prog), vp, vp, ci, vp])


def _prog(spec, M, tensors, grads=None):
    """ctypes program of `spec` over the tensors `tensors` (weights / biases / inst table in spec order); grads: matching output tensors or None."""
    p = _Prog()
    p.n_layers, p.row_stride = len(spec["layers"]), spec["row_stride"]
    it = iter(range(len(tensors)))
    for l, L in enumerate(spec["layers"]):
        q = p.layer[l]
        iw = next(it)
        q.W = _lib.ptr(tensors[iw])
        q.dW = _lib.ptr(grads[iw]) if grads is not None else None
        if L["bias"]:
            ib = next(it)
            q.b = _lib.ptr(tensors[ib])
            q.db = _lib.ptr(grads[ib]) if grads is not None else None
        q.in_dim, q.out_dim, q.src_col, q.dst_col, q.relu = L["in_dim"], L["out_dim"], L["src"], L["dst"], int(L["relu"])
    t = spec.get("time")
    if t is not None:
        p.frame_id, p.vstart, p.vidlen = _lib.ptr(t["frame_id"]), _lib.ptr(t["vstart"]), _lib.ptr(t["vidlen"])
        p.vid = _lib.ptr(t["vid"]) if t["vid"] is not None else None
        p.max_ts, p.time_scale, p.n_freq, p.four_col = float(t["max_ts"]), float(t["time_scale"]), int(t["n_freq"]), int(t["four_col"])
        if t["inst_dim"] > 0:
            ii = next(it)
            p.inst_W = _lib.ptr(tensors[ii])
            p.d_inst_W = _lib.ptr(grads[ii]) if grads is not None else None
        p.inst_col, p.inst_dim, p.inst_rows = int(t["inst_col"]), int(t["inst_dim"]), int(t["inst_rows"])
    return p


class _Run(Function):
    """(spec, n_inputs, *inputs, *params) -> the output column ranges.  spec is a plain dict (shapes, columns, frame tables)."""

    @staticmethod
    def forward(ctx, spec, n_in, *args):
        ins, params = args[:n_in], [a.detach().contiguous() for a in args[n_in:]]
        M, rs = spec["M"], spec["row_stride"]
        dev = params[0].device
        _lib.require_device(*params)
        work = torch.empty(M, rs, device=dev)
        for (col, width), x in zip(spec["inputs"], ins):
            work[:, col:col + width] = x.detach().reshape(M, width)
        p = _prog(spec, M, params)
        _lib.check(_lib.lib().lab4d_rowmlp_forward(ctypes.byref(p), _lib.ptr(work), M, _lib.stream()), "rowmlp_forward")
        ctx.spec, ctx.n_in = spec, n_in
        ctx.save_for_backward(work, *params)
        return tuple(work[:, c:c + w].clone() for c, w in spec["outs"])

    @staticmethod
    @once_differentiable
    def backward(ctx, *gouts):
        spec, n_in = ctx.spec, ctx.n_in
        work, params = ctx.saved_tensors[0], list(ctx.saved_tensors[1:])
        M, rs = spec["M"], spec["row_stride"]
        gwork = torch.zeros(M, rs, device=work.device)
        for (c, w), g in zip(spec["outs"], gouts):
            if g is not None:
                gwork[:, c:c + w] = g.reshape(M, w)
        grads = [torch.empty_like(t) for t in params]
        p = _prog(spec, M, params, grads)
        _lib.check(_lib.lib().lab4d_rowmlp_backward(ctypes.byref(p), _lib.ptr(work), _lib.ptr(gwork), M, _lib.stream()), "rowmlp_backward")
        gin = [gwork[:, c:c + w].clone() for c, w in spec["inputs"]]
        return (None, None) + tuple(gin) + tuple(grads)


def run(layers, M, outs, time=None, inputs=(), row_stride=None):
    """layers: [{"W": (out,in) tensor, "b": (out) tensor | None, "src": col, "dst": col, "relu": bool}] in execution order;
    outs: [(col, width)] column ranges to return; inputs: [((col, width), tensor (M, width))] written into the strip before the launch;
    time: {"frame_id" (M) int64, "vstart", "vidlen", "vid" (N) int64 tables, "max_ts", "time_scale", "n_freq", "four_col", "inst_W" (rows, C) | None,
    "inst_col"} = the TimeEmbedding prologue.  Returns one (M, width) tensor per entry of outs."""
    if len(layers) < 1 or len(layers) > MAX_LAYERS:
        raise RuntimeError("rowmlp.run: %d layers (1..%d)" % (len(layers), MAX_LAYERS))
    spec_layers, params = [], []
    end = 0
    for L in layers:
        W = L["W"]
        if W.dtype != torch.float32 or W.dim() != 2:
            raise RuntimeError("rowmlp.run: weights must be (out, in) fp32")
        spec_layers.append({"in_dim": W.shape[1], "out_dim": W.shape[0], "src": int(L["src"]), "dst": int(L["dst"]), "relu": bool(L.get("relu", False)),
                            "bias": L.get("b") is not None})
        params.append(W)
        if L.get("b") is not None:
            params.append(L["b"])
        end = max(end, L["src"] + W.shape[1], L["dst"] + W.shape[0])
    spec = {"layers": spec_layers, "M": int(M), "outs": [(int(c), int(w)) for c, w in outs], "inputs": [(int(c), int(w)) for (c, w), _ in inputs]}
    if time is not None:
        iw = time.get("inst_W")
        t = {"frame_id": time["frame_id"].contiguous(), "vstart": time["vstart"], "vidlen": time["vidlen"], "vid": time.get("vid"),
             "max_ts": time["max_ts"], "time_scale": time.get("time_scale", 1.0), "n_freq": max(int(time["n_freq"]), 0), "four_col": time["four_col"],
             "inst_dim": 0 if iw is None else iw.shape[1], "inst_rows": 0 if iw is None else iw.shape[0], "inst_col": time.get("inst_col", 0)}
        for k in ("frame_id", "vstart", "vidlen", "vid"):
            if t[k] is not None and t[k].dtype != torch.int64:
                raise RuntimeError("rowmlp.run: time table %r must be int64" % k)
        _lib.require_device(t["frame_id"], t["vstart"], t["vidlen"])
        spec["time"] = t
        end = max(end, t["four_col"] + 2 * t["n_freq"] + 1, t["inst_col"] + t["inst_dim"])
        if iw is not None:
            params.append(iw)
    for c, w in spec["outs"] + spec["inputs"]:
        end = max(end, c + w)
    spec["row_stride"] = int(row_stride) if row_stride is not None else (end + 3) // 4 * 4
    return _Run.apply(spec, len(inputs), *[x for _, x in inputs], *params)
