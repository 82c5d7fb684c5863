"""Weights-stationary chain kernels (csrc/mlp_kernels_ws.hpp) against the wave-resident ones (csrc/mlp_kernels.hpp), launch by launch through the C ABI:
every stored buffer (embedding, activations, ReLU sign words, dZ, ext gradient) and the outputs compared bit for bit, d_x to fp32 rounding (its
partial sums are ordered differently); then both families timed at a bench-sized launch.
  python tools/ws_compare.py [--time S] [--nets fg_base,fg_color,dense,dense6] [--json out.json]
Packed weights are random bf16 blocks (the kernels' own layout; W and W^T need not be transposes of each other for a kernel-vs-kernel comparison)."""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lab4d_amd import _lib, mlp  # noqa: E402

NETS = {"fg_base": mlp.NET_FG_BASE, "fg_color": mlp.NET_FG_COLOR, "dense": mlp.NET_DENSE, "dense6": mlp.NET_DENSE6}
BF = mlp.PREC_BF16


def make_case(net, S, spf, seed, with_freq_w=False, train=True, want_dx=True, dx_only=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    d = mlp.describe(net)
    NL = d.n_layers
    S_pad = mlp.s_pad_of(S)
    M = (S + spf - 1) // spf
    c = {"net": net, "S": S, "S_pad": S_pad, "spf": spf, "NL": NL, "d": d, "train": train, "want_dx": want_dx, "dx_only": dx_only}
    c["x"] = (torch.rand(S, 3, device="cuda", generator=g) * 0.6 - 0.3).contiguous()
    c["freq_w"] = torch.rand(d.n_freq, device="cuda", generator=g).contiguous() if with_freq_w else None
    c["W"], c["WT"], c["bias"], c["pf"] = [], [], [], []
    for l in range(NL):
        L = d.layers[l]
        K = L.ke + L.kin
        sc = 1.5 / (K ** 0.5)
        c["W"].append((torch.randn(L.mout_pad * K, device="cuda", generator=g) * sc).to(torch.bfloat16).contiguous())
        c["WT"].append((torch.randn(L.mout_pad * K, device="cuda", generator=g) * (1.5 / (L.mout_pad ** 0.5))).to(torch.bfloat16).contiguous())
        c["bias"].append((torch.randn(L.mout_pad, device="cuda", generator=g) * 0.1).contiguous())
        c["pf"].append((torch.randn(M, L.mout_pad, device="cuda", generator=g) * 0.1).contiguous() if L.pf_bias else None)
    need_ext = any(d.layers[l].add_ext for l in range(NL))
    le = [l for l in range(NL) if d.layers[l].add_ext]
    c["ext"] = (torch.randn(mlp.buf_numel(d.layers[le[0]].mout_pad, S_pad), device="cuda", generator=g) * 0.3).to(torch.bfloat16).contiguous() if need_ext else None
    lg = [l for l in range(NL) if d.layers[l].ext_grad]
    c["ext_gin"] = (torch.randn(mlp.buf_numel(d.layers[lg[0]].mout_pad, S_pad), device="cuda", generator=g) * 0.05).to(torch.bfloat16).contiguous() if lg else None
    c["export"] = lg[0] if lg else -1
    c["d_out"] = (torch.randn(S, d.c_out, device="cuda", generator=g)).contiguous()
    return c


def run_fwd(c, ws):
    os.environ["LAB4D_WS"] = "1" if ws else "0"
    d, NL, S, S_pad = c["d"], c["NL"], c["S"], c["S_pad"]
    a = mlp.FwdArgs()
    a.net, a.precision, a.S, a.S_pad, a.ld, a.spf = c["net"], BF, S, S_pad, S_pad, c["spf"]
    a.x = _lib.dp(c["x"])
    if c["freq_w"] is not None:
        a.freq_w = _lib.dp(c["freq_w"])
    r = {"act": [None] * NL, "mask": [None] * NL, "emb": None}
    for l in range(NL):
        L = d.layers[l]
        a.W[l] = _lib.dp(c["W"][l])
        a.bias[l] = _lib.dp(c["bias"][l])
        if L.pf_bias:
            a.pf_bias[l] = _lib.dp(c["pf"][l])
        if ((c["train"] and l + 1 < NL) or l == c["export"]) and not c["dx_only"]:
            r["act"][l] = torch.zeros(mlp.buf_numel(L.mout_pad, S_pad), dtype=torch.bfloat16, device="cuda")
            a.act[l] = _lib.dp(r["act"][l])
        if c["train"] and L.relu and l + 1 < NL:
            r["mask"][l] = torch.zeros((S_pad // 64) * (L.mout_pad // 32) * 64, dtype=torch.int32, device="cuda")
            a.mask[l] = _lib.dp(r["mask"][l])
    if c["train"]:
        r["emb"] = torch.zeros(mlp.buf_numel(d.ke, S_pad), dtype=torch.bfloat16, device="cuda")
        a.emb = _lib.dp(r["emb"])
    if c["ext"] is not None:
        a.ext = _lib.dp(c["ext"])
    r["out"] = torch.zeros(S, d.c_out, device="cuda")
    a.out = _lib.dp(r["out"])
    r["args"] = a
    _lib.check(_lib.lib().lab4d_mlp_forward(ctypes.byref(a), _lib.stream()), "mlp_forward")
    torch.cuda.synchronize()
    return r


def run_bwd(c, f, ws):
    os.environ["LAB4D_WS"] = "1" if ws else "0"
    d, NL, S, S_pad = c["d"], c["NL"], c["S"], c["S_pad"]
    a = mlp.BwdArgs()
    a.net, a.precision, a.S, a.S_pad, a.ld, a.spf = c["net"], BF, S, S_pad, S_pad, c["spf"]
    r = {"dz": [None] * NL}
    for l in range(NL):
        L = d.layers[l]
        a.WT[l] = _lib.dp(c["WT"][l])
        if f["act"][l] is not None:
            a.act[l] = _lib.dp(f["act"][l])
        if f["mask"][l] is not None:
            a.mask[l] = _lib.dp(f["mask"][l])
        if not c["dx_only"]:
            r["dz"][l] = torch.zeros(mlp.buf_numel(L.mout_pad, S_pad), dtype=torch.bfloat16, device="cuda")
            a.dz[l] = _lib.dp(r["dz"][l])
    if c["ext_gin"] is not None:
        a.ext_gin = _lib.dp(c["ext_gin"])
    a.emb = _lib.dp(f["emb"])
    r["ext_gout"] = None
    if c["ext"] is not None:
        a.ext = _lib.dp(c["ext"])
        r["ext_gout"] = torch.zeros_like(c["ext"])
        a.ext_gout = _lib.dp(r["ext_gout"])
    a.d_out = _lib.dp(c["d_out"])
    r["d_x"] = None
    if c["want_dx"]:
        r["d_x"] = torch.zeros(S, 3, device="cuda")
        a.d_x = _lib.dp(r["d_x"])
    r["args"] = a
    _lib.check(_lib.lib().lab4d_mlp_backward(ctypes.byref(a), _lib.stream()), "mlp_backward")
    torch.cuda.synchronize()
    return r


def run_tangent(c, f, ws):
    """lab4d_mlp_forward_tangent (the eikonal term's forward): raw (S, ke) tangent input, the primal's sign words as input, tangent activations / embedding / output written"""
    os.environ["LAB4D_WS"] = "1" if ws else "0"
    d, NL, S, S_pad = c["d"], c["NL"], c["S"], c["S_pad"]
    a = mlp.FwdArgs()
    a.net, a.precision, a.S, a.S_pad, a.ld, a.spf = c["net"], BF, S, S_pad, S_pad, c["spf"]
    a.x = _lib.dp(c["x_tan"])
    r = {"act": [None] * NL, "mask": [None] * NL}
    for l in range(NL):
        L = d.layers[l]
        a.W[l] = _lib.dp(c["W"][l])
        if f["mask"][l] is not None:
            a.mask[l] = _lib.dp(f["mask"][l])
        if l + 1 < NL:
            r["act"][l] = torch.zeros(mlp.buf_numel(L.mout_pad, S_pad), dtype=torch.bfloat16, device="cuda")
            a.act[l] = _lib.dp(r["act"][l])
    r["emb"] = torch.zeros(mlp.buf_numel(d.ke, S_pad), dtype=torch.bfloat16, device="cuda")
    a.emb = _lib.dp(r["emb"])
    r["out"] = torch.zeros(S, d.c_out, device="cuda")
    a.out = _lib.dp(r["out"])
    r["args"] = a
    _lib.check(_lib.lib().lab4d_mlp_forward_tangent(ctypes.byref(a), _lib.stream()), "mlp_forward_tangent")
    torch.cuda.synchronize()
    return r


def cmp_bits(name, x, y, F=None, report=None):
    """bit comparison of two buffers; F: feature rows of a blocked [64-sample block][feature][64] buffer (decodes the first mismatches)"""
    if x is None and y is None:
        return True
    xi = x.view(torch.int16) if x.dtype == torch.bfloat16 else x.view(torch.int32)
    yi = y.view(torch.int16) if y.dtype == torch.bfloat16 else y.view(torch.int32)
    ne = (xi != yi)
    nbad = int(ne.sum())
    row = {"buffer": name, "elements": xi.numel(), "mismatches": nbad}
    if nbad:
        idx = torch.nonzero(ne.view(-1))[:6, 0].tolist()
        where = []
        for i in idx:
            if F is not None:
                bs = F * 64 + 128
                where.append({"block": i // bs, "feature": (i % bs) // 64, "sample_in_block": (i % bs) % 64, "a": float(x.view(-1)[i].float()) if x.dtype != torch.int32 else int(xi.view(-1)[i]),
                              "b": float(y.view(-1)[i].float()) if y.dtype != torch.int32 else int(yi.view(-1)[i])})
            else:
                where.append({"index": i, "a": float(x.view(-1)[i]) if x.dtype != torch.int32 else int(xi.view(-1)[i]), "b": float(y.view(-1)[i]) if y.dtype != torch.int32 else int(yi.view(-1)[i])})
        row["first"] = where
        if x.dtype != torch.int32:
            row["max_abs_diff"] = float((x.float() - y.float()).abs().max())
        # which blocks / features are affected (pattern of the bug)
        if F is not None:
            flat = torch.nonzero(ne.view(-1))[:, 0]
            bs = F * 64 + 128
            row["blocks_hit"] = torch.unique(flat // bs)[:12].tolist()
            row["features_hit"] = torch.unique((flat % bs) // 64)[:40].tolist()
    if report is not None:
        report.append(row)
    print(("OK   " if nbad == 0 else "DIFF ") + json.dumps(row)[:600], flush=True)
    return nbad == 0


def compare(c, tag, report):
    d, NL = c["d"], c["NL"]
    ok = True
    f0, f1 = run_fwd(c, False), run_fwd(c, True)
    rep = []
    if c["train"]:
        ok &= cmp_bits("emb", f0["emb"], f1["emb"], d.ke, rep)
    for l in range(NL):
        L = d.layers[l]
        if f0["act"][l] is not None:
            ok &= cmp_bits("act[%d]" % l, f0["act"][l], f1["act"][l], L.mout_pad, rep)
        if f0["mask"][l] is not None:
            ok &= cmp_bits("mask[%d]" % l, f0["mask"][l], f1["mask"][l], None, rep)
    ok &= cmp_bits("out", f0["out"], f1["out"], None, rep)
    if c.get("tangent"):
        g = torch.Generator(device="cuda").manual_seed(99)
        c["x_tan"] = torch.randn(c["S"], d.ke, device="cuda", generator=g).contiguous()
        t0, t1 = run_tangent(c, f0, False), run_tangent(c, f0, True)
        ok &= cmp_bits("tangent emb", t0["emb"], t1["emb"], d.ke, rep)
        for l in range(NL - 1):
            ok &= cmp_bits("tangent act[%d]" % l, t0["act"][l], t1["act"][l], d.layers[l].mout_pad, rep)
        ok &= cmp_bits("tangent out", t0["out"], t1["out"], None, rep)
    if c["train"]:
        # backward of both families on the SAME (wave-resident) forward state
        b0, b1 = run_bwd(c, f0, False), run_bwd(c, f0, True)
        for l in range(NL - 1, -1, -1):
            if b0["dz"][l] is not None:
                ok &= cmp_bits("dz[%d]" % l, b0["dz"][l], b1["dz"][l], d.layers[l].mout_pad, rep)
        if b0["ext_gout"] is not None:
            le = [l for l in range(NL) if d.layers[l].add_ext][0]
            ok &= cmp_bits("ext_gout", b0["ext_gout"], b1["ext_gout"], d.layers[le].mout_pad, rep)
        if c["want_dx"]:
            err = float((b0["d_x"] - b1["d_x"]).abs().max())
            ref = float(b0["d_x"].abs().max())
            dx_ok = err <= 2e-5 * ref
            rep.append({"buffer": "d_x", "max_abs_diff": err, "max_abs": ref, "ok": dx_ok})
            print(("OK   " if dx_ok else "DIFF ") + "d_x max|diff| %.3e of max %.3e" % (err, ref), flush=True)
            ok &= dx_ok
    report.append({"case": tag, "ok": bool(ok), "buffers": rep})
    return ok


def time_case(net, S, report, dx_only=False):
    c = make_case(net, S, S // 2, 7, dx_only=dx_only)
    out = {"net": mlp.NET_NAMES[net], "S": S, "dx_only": dx_only}
    for ws in (False, True):
        f = run_fwd(c, ws)
        b = run_bwd(c, f, ws)
        os.environ["LAB4D_WS"] = "1" if ws else "0"
        for what, fn, args in (("fwd", _lib.lib().lab4d_mlp_forward, f["args"]), ("bwd", _lib.lib().lab4d_mlp_backward, b["args"])):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 5
            fn(ctypes.byref(args), _lib.stream())
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n):
                fn(ctypes.byref(args), _lib.stream())
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            tf = 2.0 * S * mlp.NET_MACS[net] / (ms * 1e-3) / 1e12
            out["%s_%s_ms" % (what, "ws" if ws else "wave")] = round(ms, 3)
            out["%s_%s_frac_of_bf16_mfma_peak" % (what, "ws" if ws else "wave")] = round(tf / 2500.0, 4)
        if ws and os.environ.get("LAB4D_WS_TRACE_PRINT"):
            t = f["out"].view(-1)[:64].tolist()
            ntile = (c["S_pad"] // 128 + 255) // 256
            names = ["block_wait", "loop0", "epi0", "loop1", "epi1", "barrier", "posenc", "vmcnt@entry"]
            out["trace_cycles_per_tile"] = {"wave%d" % wv: {names[i]: round(t[8 * wv + i] / ntile) for i in range(8)} for wv in range(8)}
        del f, b
        torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)
    report.append(out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", type=int, default=0)
    ap.add_argument("--nets", default="fg_base,fg_color,dense,dense6")
    ap.add_argument("--json", default=None)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--time-dx-only", action="store_true")
    a = ap.parse_args()
    _lib.lib()
    report, timing = [], []
    all_ok = True
    for name in a.nets.split(","):
        net = NETS[name]
        cases = [(1000, 300, False, True, True, False), (128 * 37 + 77, 1000, True, True, True, False), (4096, 2048, False, True, False, False), (700, 128, False, False, True, False)]
        if name == "fg_base":
            cases.append((128 * 9 + 37, 700, False, True, True, False))  # + the tangent-mode forward of the eikonal term on this case's sign words (flag set below)
            cases.append((128 * 21 + 5, 512, True, True, True, True))  # point-gradient-only mode (sign words + embedding only, no dZ): the sdf basefields
        if a.quick:
            cases = cases[:1]
        for i, (S, spf, fw, train, dx, dxo) in enumerate(cases):
            tag = "%s S=%d spf=%d freq_w=%s train=%s dx=%s dx_only=%s" % (name, S, spf, fw, train, dx, dxo)
            print("== " + tag, flush=True)
            try:
                cs = make_case(net, S, spf, 11 + i, fw, train, dx, dxo)
                cs["tangent"] = (name == "fg_base" and S == 128 * 9 + 37)
                all_ok &= compare(cs, tag, report)
            except Exception as e:  # a failing launch must not hide the other cases
                print("EXC  " + repr(e), flush=True)
                report.append({"case": tag, "ok": False, "exception": repr(e)})
                all_ok = False
    if a.time:
        for name in a.nets.split(","):
            time_case(NETS[name], a.time, timing)
            if name == "fg_base" and a.time_dx_only:
                time_case(NETS[name], a.time, timing, dx_only=True)
    if a.json:
        json.dump({"all_bit_equal": bool(all_ok), "cases": report, "timing": timing}, open(a.json, "w"), indent=1)
    print("ALL_OK" if all_ok else "SOME_DIFF")
    sys.exit(0 if all_ok else 1)
