"""Which loss term carries a gradient tensor's deviation?  Device (fp32 chains) against the oracle in float64 on a comp fixture, term by term.
   python tools/diag/grad_terms.py [comp_bench] [tensor names ...]       (TEST INFRASTRUCTURE: imports the oracle)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from lab4d_amd import deformable as DF, mlp, synthetic
from oracle import lab4d_oracle as O
from fixture_utils import bg_weights, fg_weights, leaf, rays_and_targets

name = sys.argv[1] if len(sys.argv) > 1 else "comp_bench"
tensors = sys.argv[2:] or ["basefield.linear_3.0.weight", "basefield.linear_4.0.weight", "basefield.linear_5.0.weight", "basefield.linear_6.0.weight", "basefield.linear_4.0.bias"]
g = torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"), weights_only=False)
meta = g["meta"]
hxy0, batch0 = rays_and_targets(g)
torch.set_num_threads(min(os.cpu_count(), 32))


def to(x, dt):
    if torch.is_tensor(x):
        return x.to(dt) if x.dtype.is_floating_point else x
    if isinstance(x, tuple):
        return tuple(to(t, dt) for t in x)
    if isinstance(x, dict):
        return {k: to(v, dt) for k, v in x.items()}
    return x


def oracle(dt):
    Pf = {k: (to(v, dt).clone().requires_grad_(True) if v.dtype.is_floating_point and k != "aabb" else to(v, dt)) for k, v in fg_weights(meta).items()}
    Pb = {k: to(v, dt).clone().requires_grad_(True) for k, v in bg_weights(meta).items()}
    frf = synthetic.add_codes(to(dict(g["frames_fg"]), dt), Pf)
    batch = to(batch0, dt)
    frf["feature"] = batch["feature"]
    frb = synthetic.add_bg_codes(to(dict(g["frames_bg"]), dt), Pb)
    res = O.render_train_comp(Pf, frf, Pb, frb, to(hxy0, dt), g["rng"], flow_thresh=meta["flow_thresh"], n_depth=meta["D"])
    losses = O.recon_losses_comp(res, batch, meta["res"], O.DEFAULT_LOSS_WT)
    return {k: torch.autograd.grad(v, [Pf[t] for t in tensors], retain_graph=True, allow_unused=True) for k, v in losses.items()}


def device(prec):
    Pf, Pb = leaf(fg_weights(meta), "cuda"), {k: v.cuda().clone().requires_grad_(True) for k, v in bg_weights(meta).items()}
    batch = synthetic.to_device(batch0, "cuda")
    frf = synthetic.add_codes(synthetic.to_device(dict(g["frames_fg"]), "cuda"), Pf)
    frf["feature"] = batch["feature"]
    frb = synthetic.add_bg_codes(synthetic.to_device(dict(g["frames_bg"]), "cuda"), Pb)
    out = {}
    for k in TERMS:  # the chain kernels release their stored activations in backward: one render per term
        res = DF.render_train_comp(Pf, frf, Pb, frb, hxy0.cuda(), synthetic.to_device(g["rng"], "cuda"), flow_thresh=meta["flow_thresh"], n_depth=meta["D"], prec=prec)
        losses = DF.losses_comp(res, batch, meta["res"], DF.DEFAULT_LOSS_WT)
        out[k] = torch.autograd.grad(losses[k], [Pf[t] for t in tensors], allow_unused=True)
    return out


o64, o32 = oracle(torch.float64), oracle(torch.float32)
TERMS = list(o64.keys())
d32 = device(mlp.PREC_F32)
out = {}
for ti, t in enumerate(tensors):
    tot64 = sum(o64[k][ti].double() for k in o64 if o64[k][ti] is not None)
    rows = {}
    for k in o64:
        if o64[k][ti] is None:
            continue
        r = o64[k][ti].double()
        dev = d32[k][ti].double().cpu() if d32[k][ti] is not None else torch.zeros_like(r)
        rows[k] = {"share_of_total_norm": float(r.norm() / tot64.norm()), "device_err_over_total_norm": float((dev - r).norm() / tot64.norm()),
                   "oracle_fp32_err_over_total_norm": float((o32[k][ti].double() - r).norm() / tot64.norm())}
    out[t] = rows
    print(t)
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["device_err_over_total_norm"]):
        print("   %-20s share %.3f  device err %.2e  oracle-fp32 err %.2e" % (k, v["share_of_total_norm"], v["device_err_over_total_norm"], v["oracle_fp32_err_over_total_norm"]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_grad_terms_%s.json" % name), "w"), indent=1)
