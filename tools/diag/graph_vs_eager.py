"""Diagnostic: one training chunk of the bench (same shapes) run eagerly and as a captured + replayed hipGraph; every per-sample field's
gradient, every prologue leaf gradient and every parameter gradient is compared between the two.  Prints the tensors that differ."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=128)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--poison", action="store_true")
    ap.add_argument("--sync-capture", action="store_true")
    a = ap.parse_args()
    if a.poison:
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True
    dev = torch.device("cuda", 0)
    from lab4d_amd import _lib, mlp
    from lab4d_amd import deformable as DF, render_utils as RU
    from lab4d_amd.optim import FlatAdamW
    _lib.lib()
    prec = mlp.PREC_BF16
    P, fr = bench.make_problem(a.res, dev)
    names = [k for k, v in P.items() if v.dtype.is_floating_point and v.requires_grad]
    params = [P[k] for k in names]
    opt = FlatAdamW(params, lr=5e-4)
    mlp.FUSED_GRAD_ACCUM = True
    rows = list(range(0, a.res, a.res // a.rows))[: a.rows]
    hxy, batch = bench.chunk_inputs(a.res, None, rows, dev, seed=100)
    batch["hxy"] = hxy
    M, N0 = hxy.shape[:2]
    gen = torch.Generator(device=dev).manual_seed(1234)
    rng = bench.draw_rng(M, N0, M * N0 * a.spp, dev, gen)
    prologue = DF.FramePrologue(P, fr)
    frs = prologue.refresh()
    prologue.outs = None

    saved = {}
    fwd = {}

    def chunk(tag):
        f = dict(frs)
        f["feature"] = batch["feature"]
        fd, deltas, aux = DF.query_field_train(P, f, hxy, rng, float(a.res), a.spp, None, prec)
        for k, v in fd.items():
            if v.requires_grad:
                v.register_hook(lambda g, k=k: saved.__setitem__((tag, "d/" + k), g.detach().clone() if tag == "eager" else g))
        if tag == "graph":
            fwd.update({"fd/" + k: v for k, v in fd.items()})
            fwd["deltas"] = deltas
        rendered = RU.render_pixel(fd, deltas)
        if tag == "graph":
            fwd.update({"r/" + k: v for k, v in rendered.items()})
            fwd.update({"aux/" + k: v for k, v in aux.items()})
        aux_fg = dict(aux)
        aux_fg.update(rendered)
        rendered = dict(rendered)
        rendered["xyz_matches"], rendered["xyz_reproj"] = aux["xyz_matches"], aux["xyz_reproj"]
        for k, v in rendered.items():
            if v.requires_grad:
                v.register_hook(lambda g, k=k: saved.__setitem__((tag, "r/" + k), g.detach().clone() if tag == "eager" else g))
        losses = DF.losses_fg({"rendered": rendered, "aux_dict": {"fg": aux_fg}}, batch, a.res, DF.DEFAULT_LOSS_WT)
        if tag == "graph":
            fwd["lossvec"] = losses.vec
        losses.total.backward()
        return losses.vec.detach()

    def snapshot(tag):
        torch.cuda.synchronize()
        out = {"p/" + n: q.grad.detach().clone() for n, q in zip(names, params)}
        out.update({"leaf/" + k: v.grad.detach().clone() for k, v in prologue.leaves.items() if v.grad is not None})
        return out

    def zero():
        opt.zero_grad()
        prologue.zero_grad()

    zero()
    v_e = chunk("eager").clone()
    g_e = snapshot("eager")
    for k in list(saved):
        saved[k] = saved[k].cpu()
    zero()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    # capture exactly as bench.py does
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chunk("warm")
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for k in [k for k in saved if k[0] == "warm"]:
        del saved[k]
    torch.cuda.empty_cache()
    zero()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        v_g = chunk("graph")
    zero()
    fwd0 = None
    for rep in range(3):
        graph.replay()
        g_g = snapshot("graph")
        if fwd0 is None:
            fwd0 = {k: v.detach().clone() for k, v in fwd.items()}
            bwd0 = {k: v.detach().clone() for k, v in saved.items() if k[0] == "graph"}
        else:
            print("replay %d vs replay 0, forward tensors that changed:" % rep)
            for k, v in fwd.items():
                d = (v.detach() != fwd0[k])
                if bool(d.any()):
                    print("   fwd", k, tuple(v.shape), "changed elements:", int(d.sum()), "max |new|", float(v.detach()[d].abs().max()))
            for k, v in saved.items():
                if k[0] == "graph":
                    d = (v.detach() != bwd0[k])
                    if bool(d.any()):
                        print("   bwd", k[1], tuple(v.shape), "changed elements:", int(d.sum()))
        print("replay %d: loss eager %.6g graph %.6g" % (rep, float(v_e[12]), float(v_g[12])))
        bad = []
        for k, e in g_e.items():
            g = g_g[k]
            d = float((g - e).abs().max())
            s = float(e.abs().max())
            if not (d <= 1e-3 * s + 1e-12):
                bad.append((k, d, s))
        for (tag, k), e in saved.items():
            if tag != "eager":
                continue
            g = saved[("graph", k)].detach().cpu()
            d = float((g - e).abs().max())
            s = float(e.abs().max())
            if not (d <= 1e-3 * s + 1e-12):
                nb = int(((g - e).abs() > 1e-3 * s + 1e-12).sum())
                bad.append((k, d, s, nb, e.numel()))
        print("  differing tensors (name, max |diff|, max |eager|[, count, numel]):")
        for b in bad:
            print("   ", b)
        zero()


if __name__ == "__main__":
    main()
