import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from lab4d_amd import synthetic, deformable as DF
g = torch.load(os.path.join(sys.path[0], "tests/golden/eval_small.pt"), weights_only=False)
P = synthetic.make_weights(g["meta"]["seed"], sdf_bias=g["meta"].get("sdf_bias"))
Pd = synthetic.to_device(P, "cuda")
fr = synthetic.add_codes(synthetic.to_device(dict(g["frames"]), "cuda"), Pd)
out = DF.render_eval(Pd, fr, g["hxy"].cuda(), n_depth=g["meta"]["D"])
print("inds mismatch", (out["debug"]["inds"].cpu() != g["inds"]).float().mean().item(), "valid mismatch", (out["debug"]["valid"].cpu() != g["valid"]).float().mean().item())
for k, v in g["rendered"].items():
    a = out["rendered"][k].cpu()
    print(k, float((a - v).abs().max() / (v.abs().max() + 1e-12)))
