"""fp32 noise floor of the REFERENCE arithmetic on every parity fixture: the oracle (oracle/lab4d_oracle.py, the CPU restatement pinned to the
reference) evaluated in float32 and in float64 on the same inputs.  The difference is what ANY fp32 implementation with a different order of
accumulation can differ from the reference by (discrete events included: a ReLU unit or an importance-sampling bin that flips between the
two precisions); the device parity bounds of tests/test_gpu_field.py that exceed north_star's 1e-4 are held against it.

    python tests/measure_fp32_noise_floor.py [case ...]      -> tests/golden/fp32_noise_floor.json (merged)

TEST INFRASTRUCTURE ONLY.  Metrics: rendered / per-sample / loss entries = max |a - b| / max |b|; gradients = relative L2 (the metrics of the tests)."""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from lab4d_amd import synthetic  # noqa: E402
from oracle import lab4d_oracle as O  # noqa: E402

OUT = os.path.join(HERE, "golden", "fp32_noise_floor.json")
TRAIN = ["train_small", "train_alpha", "train_multi", "train_multi10", "train_compmotion", "train_human", "train_rigid", "train_dense", "train_c1", "train_bench",
         "train_multi10_bench", "train_bench_w1"]
EVAL = ["eval_small", "eval_rigid", "eval_dense"]
EVAL_BENCH = ["eval_bench", "eval_bench_w1"]
COMP_EVAL_BENCH = ["comp_eval_bench"]  # the comp configuration's eval path at configs[2]'s shape  # round 5: the eval path at the bench size (band by band, like the fixture)
COMP = ["comp_train", "comp_bench"]  # field_type "comp": fg + bg composite (round 4)


def to(x, dt):
    if torch.is_tensor(x):
        return x.to(dt) if x.dtype.is_floating_point else x
    if isinstance(x, tuple):
        return tuple(to(t, dt) for t in x)
    if isinstance(x, dict):
        return {k: to(v, dt) for k, v in x.items()}
    return x


def weights_of(meta):
    sys.path.insert(0, HERE)
    from fixture_utils import fg_weights
    return fg_weights(meta)


def relmax(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300))


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def grad_l2(fix, key, a, b):
    """Relative L2 of a gradient tensor IN THE METRIC OF THE TESTS: a fixture stores a big gradient as every stride-th entry (make_golden.compress_grad) and the
    device tests compare that subsample -- whose relative error is not the full tensor's (comp_bench, basefield.linear_4.0.weight: 8.7e-5 over all 65,536 entries,
    1.4e-3 over the 1,024 stored ones, which sit in four input columns with small gradients).  The floor is taken on the same entries."""
    ref = fix.get("grads", {}).get(key)
    if ref is not None and "stride" in ref:
        return rel_l2(a.flatten()[:: ref["stride"]], b.flatten()[:: ref["stride"]])
    return rel_l2(a, b)


def train_case(name):
    g = torch.load(os.path.join(HERE, "golden", name + ".pt"), weights_only=False)
    meta = g["meta"]
    if "hxy" in g:
        hxy0, batch0 = g["hxy"], g["batch"]
    else:
        hxy0 = synthetic.make_rays(meta["res"], meta["M"], rows=meta.get("rows"))
        batch0 = synthetic.make_targets(meta["seed"] + 3, meta["M"], hxy0.shape[1], meta["res"], hxy0)

    def run(dt):
        P = {k: (to(v, dt).clone().requires_grad_(True) if v.dtype.is_floating_point and k != "aabb" else to(v, dt)) for k, v in weights_of(meta).items()}
        fr = synthetic.add_codes(to(dict(g["frames"]), dt), P)
        batch = to(batch0, dt)
        fr["feature"] = batch["feature"]
        res = O.render_train(P, fr, to(hxy0, dt), g["rng"], flow_thresh=meta["flow_thresh"], n_depth=meta["D"], alpha=meta["alpha"])
        if "hxy" in g:  # the small fixtures also pin the per-sample fields
            with torch.no_grad():
                res["feat_dict"], res["deltas"], _ = O.query_field_train(P, fr, to(hxy0, dt), g["rng"], flow_thresh=meta["flow_thresh"], n_depth=meta["D"],
                                                                        alpha=meta["alpha"])
        losses = O.recon_losses_fg(res, batch, meta["res"], O.DEFAULT_LOSS_WT)
        names = [k for k, v in P.items() if v.requires_grad]
        total = sum(v for v in losses.values() if bool(torch.isfinite(v)))
        gr = torch.autograd.grad(total, [P[k] for k in names], allow_unused=True)
        return res, losses, {k: v for k, v in zip(names, gr) if v is not None}

    (r32, l32, g32), (r64, l64, g64) = run(torch.float32), run(torch.float64)
    out = {}
    for k, v in r32["rendered"].items():
        out["rendered." + k] = relmax(v, r64["rendered"][k])
    for k, v in r32.get("feat_dict", {}).items():
        out["feat_dict." + k] = relmax(v, r64["feat_dict"][k])
    for k, v in r32["aux_dict"]["fg"].items():
        out["aux_fg." + k] = relmax(v, r64["aux_dict"]["fg"][k])
    for k, v in l32.items():
        if bool(torch.isfinite(v)):
            out["loss." + k] = relmax(v, l64[k])
    for k, v in g32.items():
        out["grad." + k] = grad_l2(g, k, v, g64[k])
        out["gradmax." + k] = relmax(v, g64[k])
    return out


def comp_case(name):
    """The comp training graph (fg + bg, compose_fields, comp losses) in float32 against float64: keys as tests/test_gpu_field.py reports them
    (rendered.* / aux_fg.* / aux_bg.* / loss.* / grad.fg:* / grad.bg:*)."""
    sys.path.insert(0, HERE)
    from fixture_utils import bg_weights, fg_weights, rays_and_targets
    g = torch.load(os.path.join(HERE, "golden", name + ".pt"), weights_only=False)
    meta = g["meta"]
    hxy0, batch0 = rays_and_targets(g)

    def run(dt):
        Pf = {k: (to(v, dt).clone().requires_grad_(True) if v.dtype.is_floating_point and k != "aabb" else to(v, dt)) for k, v in fg_weights(meta).items()}
        Pb = {k: to(v, dt).clone().requires_grad_(True) for k, v in bg_weights(meta).items()}
        frf = synthetic.add_codes(to(dict(g["frames_fg"]), dt), Pf)
        batch = to(batch0, dt)
        frf["feature"] = batch["feature"]
        frb = synthetic.add_bg_codes(to(dict(g["frames_bg"]), dt), Pb)
        res = O.render_train_comp(Pf, frf, Pb, frb, to(hxy0, dt), g["rng"], flow_thresh=meta["flow_thresh"], n_depth=meta["D"])
        if "bg_feat_dict" in g:  # the 12-ray fixture also pins the bg field's per-sample outputs
            with torch.no_grad():
                res["bg_feat_dict"] = O.query_field_train_bg(Pb, frb, to(hxy0, dt), g["rng"], flow_thresh=meta["flow_thresh"], n_depth=meta["D"])[0]
        losses = O.recon_losses_comp(res, batch, meta["res"], O.DEFAULT_LOSS_WT)
        names = ["fg:" + k for k, v in Pf.items() if v.requires_grad] + ["bg:" + k for k in Pb]
        gr = torch.autograd.grad(sum(losses.values()), [(Pf if n.startswith("fg:") else Pb)[n[3:]] for n in names], allow_unused=True)
        return res, losses, {k: v for k, v in zip(names, gr) if v is not None}

    (r32, l32, g32), (r64, l64, g64) = run(torch.float32), run(torch.float64)
    out = {}
    for k, v in r32["rendered"].items():
        out["rendered." + k] = relmax(v, r64["rendered"][k])
    for k, v in r32.get("bg_feat_dict", {}).items():
        out["bg_feat_dict." + k] = relmax(v, r64["bg_feat_dict"][k])
    for cat in ("fg", "bg"):
        for k, v in r32["aux_dict"][cat].items():
            if torch.is_tensor(v):
                out["aux_%s.%s" % (cat, k)] = relmax(v, r64["aux_dict"][cat][k])
    for k, v in l32.items():
        out["loss." + k] = relmax(v, l64[k])
    for k, v in g32.items():
        out["grad." + k] = grad_l2(g, k, v, g64[k])
        out["gradmax." + k] = relmax(v, g64[k])
    return out


def eval_case(name):
    g = torch.load(os.path.join(HERE, "golden", name + ".pt"), weights_only=False)
    meta = g["meta"]

    def run(dt):
        P = to(weights_of(meta), dt)
        fr = synthetic.add_codes(to(dict(g["frames"]), dt), P)
        out = O.render_eval(P, fr, to(g["hxy"], dt), n_depth=meta["D"])
        fd, _, _ = O.query_field_eval(P, fr, to(g["hxy"], dt), n_depth=meta["D"])
        return out, fd

    (o32, f32), (o64, f64) = run(torch.float32), run(torch.float64)
    out = {"index_mismatch_count": float((o32["debug"]["inds"] != o64["debug"]["inds"]).sum())}
    for k, v in o32["rendered"].items():
        out["rendered." + k] = relmax(v, o64["rendered"][k])
    for k, v in f32.items():
        out["feat_dict." + k] = relmax(v, f64[k])
    return out


def eval_bench_case(name):
    """The eval path at the bench size: the oracle in float32 against float64 band by band, on the rays the fixture stores (every stride-th), concatenated
    over the bands -- the metric of tests/test_gpu_field.py::test_eval_graph_at_the_bench_size; plus how many importance indices / mask bits the two
    precisions themselves disagree on (the discrete part of the floor)."""
    sys.path.insert(0, HERE)
    from fixture_utils import eval_bench_bands
    g = torch.load(os.path.join(HERE, "golden", name + ".pt"), weights_only=False)
    meta = g["meta"]
    st = meta["full_grid_stride"]
    acc = {32: {}, 64: {}}
    idx_mis = mask_mis = 0
    for band, hxy, _ in eval_bench_bands(g):
        outs = {}
        for bits, dt in ((32, torch.float32), (64, torch.float64)):
            P = to(weights_of(meta), dt)
            fr = synthetic.add_codes(to(dict(g["frames"]), dt), P)
            outs[bits] = O.render_eval(P, fr, to(hxy, dt), n_depth=meta["D"])
            for k, v in outs[bits]["rendered"].items():
                acc[bits].setdefault(k, []).append(v[:, ::st])
        idx_mis += int((outs[32]["debug"]["inds"] != outs[64]["debug"]["inds"]).sum())
        mask_mis += int((outs[32]["debug"]["valid"] != outs[64]["debug"]["valid"]).sum())
    out = {"index_mismatch_count": float(idx_mis), "valid_mask_mismatch_count": float(mask_mis)}
    for k in acc[32]:
        out["rendered." + k] = relmax(torch.cat(acc[32][k], 1), torch.cat(acc[64][k], 1))
    # the normal is compared opacity-weighted (see tests/test_gpu_field.py: _run_eval_bench); the fixture's mask is the float32 reference's
    m_ref = torch.cat([r["mask"] for r in g["rendered_bands"]], 1).double()
    out["rendered.normal"] = rel_l2(torch.cat(acc[32]["normal"], 1).double() * m_ref, torch.cat(acc[64]["normal"], 1) * m_ref)  # relative L2 over the stored rays
    return out


def comp_eval_bench_case(name):
    """comp_eval_bench: the oracle's comp eval render in float32 against float64, band by band, in the metric of tests/test_gpu_field.py::_run_comp_eval_bench."""
    sys.path.insert(0, HERE)
    from fixture_utils import bg_weights, eval_bench_bands, fg_weights
    g = torch.load(os.path.join(HERE, "golden", name + ".pt"), weights_only=False)
    meta = g["meta"]
    st = meta["full_grid_stride"]
    acc = {32: {}, 64: {}}
    mis = {"fg": 0, "bg": 0, "valid": 0}
    for band, hxy, _ in eval_bench_bands(g):
        outs = {}
        for bits, dt in ((32, torch.float32), (64, torch.float64)):
            Pf, Pb = to(fg_weights(meta), dt), to(bg_weights(meta), dt)
            frf = synthetic.add_codes(to(dict(g["frames_fg"]), dt), Pf)
            frb = synthetic.add_bg_codes(to(dict(g["frames_bg"]), dt), Pb)
            o = outs[bits] = O.render_eval_comp(Pf, frf, Pb, frb, to(hxy, dt), n_depth=meta["D"])
            for nm, r in (("rendered", o["rendered"]), ("fg", o["aux_dict"]["fg"]), ("bg", o["aux_dict"]["bg"])):
                for k, v in r.items():
                    acc[bits].setdefault((nm, k), []).append(v[:, ::st])
        for fld in ("fg", "bg"):
            mis[fld] += int((outs[32]["debug"][fld]["inds"] != outs[64]["debug"][fld]["inds"]).sum())
        mis["valid"] += int((outs[32]["debug"]["fg"]["valid"] != outs[64]["debug"]["fg"]["valid"]).sum())
    out = {"index_mismatch_count_fg": float(mis["fg"]), "index_mismatch_count_bg": float(mis["bg"]), "valid_mask_mismatch_count": float(mis["valid"])}
    masks = {"rendered": "rendered_bands", "fg": "rendered_fg_bands", "bg": "rendered_bg_bands"}
    for (nm, k) in acc[32]:
        a, b = torch.cat(acc[32][(nm, k)], 1), torch.cat(acc[64][(nm, k)], 1)
        if k == "normal":
            m_ref = torch.cat([r["mask"] for r in g[masks[nm]]], 1).double()
            out["%s.%s" % (nm, k)] = rel_l2(a.double() * m_ref, b * m_ref)
        else:
            out["%s.%s" % (nm, k)] = relmax(a, b)
    return out


def main(cases):
    torch.set_num_threads(os.cpu_count() or 8)
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for c in cases:
        t = time.time()
        res[c] = {k: float("%.3e" % v) for k, v in (train_case(c) if c in TRAIN else comp_case(c) if c in COMP else eval_bench_case(c) if c in EVAL_BENCH else comp_eval_bench_case(c) if c in COMP_EVAL_BENCH else eval_case(c)).items()}
        print(c, "%.1f s" % (time.time() - t), "worst:", sorted(((v, k) for k, v in res[c].items() if not k.startswith("gradmax")), reverse=True)[:3])
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:] or TRAIN + EVAL + COMP + EVAL_BENCH + COMP_EVAL_BENCH)
