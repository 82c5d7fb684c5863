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
                ("acc", i32)]


class _IO(ctypes.Structure):
    _fields_ = [("ptr", vp), ("col", i32), ("width", i32)]


MAX_IO = 4


class _Prog(ctypes.Structure):
    _fields_ = [("n_layers", i32), ("row_stride", i32), ("frame_id", vp), ("vstart", vp), ("vidlen", vp), ("vid", vp), ("inst_W", vp), ("d_inst_W", vp),
                ("max_ts", cf), ("time_scale", cf), ("n_freq", i32), ("four_col", i32), ("inst_col", i32), ("inst_dim", i32), ("inst_rows", i32),
                ("acc_inst", i32), ("n_in", i32), ("n_out", i32), ("inp", _IO * MAX_IO), ("out", _IO * MAX_IO), ("layer", _Layer * MAX_LAYERS)]


_lib.register("lab4d_rowmlp_forward", [ctypes.POINTER(_Prog), vp, ci, vp])
_lib.register("lab4d_rowmlp_backward", [ctypes.POINTER(_Prog), vp, vp, ci, vp])


def _prog(spec, tensors, ins, outs, grads=None, acc=None):
    """ctypes program of `spec` over the tensors `tensors` (weights / biases / inst table in spec order); ins / outs: the tensors (or None) bound to the
    program's input / output column ranges; grads: the gradient targets matching `tensors` (or None), acc[i]: grads[i] is accumulated into."""
    p = _Prog()
    p.n_layers, p.row_stride = len(spec["layers"]), spec["row_stride"]
    it = iter(range(len(tensors)))
    g = (lambda i: _lib.ptr(grads[i]) if grads is not None and grads[i] is not None else None)
    ac = (lambda i: bool(acc is not None and acc[i]))
    for l, L in enumerate(spec["layers"]):
        q = p.layer[l]
        iw = next(it)
        q.W, q.dW, a = _lib.ptr(tensors[iw]), g(iw), int(ac(iw))
        if L["bias"]:
            ib = next(it)
            q.b, q.db = _lib.ptr(tensors[ib]), g(ib)
            a |= 2 * int(ac(ib))
        q.in_dim, q.out_dim, q.src_col, q.dst_col, q.relu, q.acc = L["in_dim"], L["out_dim"], L["src"], L["dst"], int(L["relu"]), a
    t = spec.get("time")
    if t is not None:
        p.frame_id, p.vstart, p.vidlen = _lib.ptr(t["frame_id"]), _lib.ptr(t["vstart"]), _lib.ptr(t["vidlen"])
        p.vid = _lib.ptr(t["vid"]) if t["vid"] is not None else None
        p.max_ts, p.time_scale, p.n_freq, p.four_col = float(t["max_ts"]), float(t["time_scale"]), int(t["n_freq"]), int(t["four_col"])
        if t["inst_dim"] > 0:
            ii = next(it)
            p.inst_W, p.d_inst_W, p.acc_inst = _lib.ptr(tensors[ii]), g(ii), int(ac(ii))
        p.inst_col, p.inst_dim, p.inst_rows = int(t["inst_col"]), int(t["inst_dim"]), int(t["inst_rows"])
    p.n_in, p.n_out = len(spec["inputs"]), len(spec["outs"])
    for q, ((c, w), x) in enumerate(zip(spec["inputs"], ins)):
        p.inp[q].ptr, p.inp[q].col, p.inp[q].width = (_lib.ptr(x) if x is not None else None), c, w
    for q, ((c, w), x) in enumerate(zip(spec["outs"], outs)):
        p.out[q].ptr, p.out[q].col, p.out[q].width = (_lib.ptr(x) if x is not None else None), c, w
    return p


class _Run(Function):
    """(spec, n_inputs, *inputs, *params) -> the output column ranges as fresh (M, width) tensors.  spec is a plain dict (shapes, columns, frame tables).
    Gradients of parameters whose `.grad` is a fused-accumulation sink (lab4d_amd.mlp.FUSED_GRAD_ACCUM: views of FlatAdamW's flat buffer) are ADDED
    there by the parameter kernel and autograd is handed None for them -- no AccumulateGrad launch per parameter; the others are returned."""

    @staticmethod
    def forward(ctx, spec, n_in, *args):
        M, rs = spec["M"], spec["row_stride"]
        ins = [a.detach().reshape(M, w).contiguous().float() for a, (_, w) in zip(args[:n_in], spec["inputs"])]
        params = [a.detach().contiguous() for a in args[n_in:]]
        dev = params[0].device
        _lib.require_device(*params, *ins)
        work = torch.empty(M, rs, device=dev)
        outs = [torch.empty(M, w, device=dev) for _, w in spec["outs"]]
        p = _prog(spec, params, ins, outs)
        _lib.check(_lib.lib().lab4d_rowmlp_forward(ctypes.byref(p), _lib.ptr(work), M, _lib.stream()), "rowmlp_forward")
        ctx.spec, ctx.n_in, ctx.param_refs = spec, n_in, args[n_in:]
        ctx.save_for_backward(work, *params)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gouts):
        from . import mlp  # (the fused-accumulation switch lives there)
        spec, n_in = ctx.spec, ctx.n_in
        work, params = ctx.saved_tensors[0], list(ctx.saved_tensors[1:])
        M, rs = spec["M"], spec["row_stride"]
        dev = work.device
        gwork = torch.empty(M, rs, device=dev)  # (cleared by the chain kernel)
        gos = [None if g is None else g.reshape(M, w).contiguous().float() for g, (_, w) in zip(gouts, spec["outs"])]
        gins = [torch.empty(M, w, device=dev) if ctx.needs_input_grad[2 + i] else None for i, (_, w) in enumerate(spec["inputs"])]
        grads, acc, ret = [], [], []
        for i, (t, ref) in enumerate(zip(params, ctx.param_refs)):
            if not ctx.needs_input_grad[2 + n_in + i]:
                grads.append(None); acc.append(False); ret.append(None)
                continue
            sink = mlp._grad_sink(ref)
            if sink is not None:
                grads.append(sink); acc.append(True); ret.append(None)
            else:
                gt = torch.empty_like(t)
                grads.append(gt); acc.append(False); ret.append(gt)
        p = _prog(spec, params, gins, gos, grads, acc)
        _lib.check(_lib.lib().lab4d_rowmlp_backward(ctypes.byref(p), _lib.ptr(work), _lib.ptr(gwork), M, _lib.stream()), "rowmlp_backward")
        return (None, None) + tuple(gins) + tuple(ret)


def run(layers, M, outs, time=None, inputs=(), row_stride=None):
    """layers: [{"W": (out,in) tensor, "b": (out) tensor | None, "src": col, "dst": col, "relu": bool}] in execution order;
    outs: [(col, width)] column ranges to return; inputs: [((col, width), tensor (M, width))] written into the strip before the launch;
    time: {"frame_id" (M) int64, "vstart", "vidlen", "vid" (N) int64 tables, "max_ts", "time_scale", "n_freq", "four_col", "inst_W" (rows, C) | None,
    "inst_col"} = the TimeEmbedding prologue.  Returns one (M, width) tensor per entry of outs."""
    if len(layers) < 1 or len(layers) > MAX_LAYERS:
        raise RuntimeError("rowmlp.run: %d layers (1..%d)" % (len(layers), MAX_LAYERS))
    if len(outs) > MAX_IO or len(inputs) > MAX_IO:
        raise RuntimeError("rowmlp.run: at most %d inputs and %d outputs" % (MAX_IO, MAX_IO))
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
