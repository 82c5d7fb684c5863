"""Per-kernel timing of the chain kernels at the bench's launch size (HIP events around every launch, lab4d_amd._lib.PROF).
usage: python tools/bench_chain.py [S=4194304] [nets=base,color,...]   (LAB4D_SO_PATH selects a kernel-experiment build)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lab4d_amd import _lib, mlp, synthetic
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
which = (sys.argv[2] if len(sys.argv) > 2 else "base,color").split(",")
P = synthetic.to_device(synthetic.make_weights(0), "cuda")
for k in P:
    if P[k].dtype.is_floating_point:
        P[k].requires_grad_(True)
fr = synthetic.to_device(synthetic.add_codes(synthetic.make_frames(1, 2, 512), synthetic.make_weights(0)), "cuda")
x = (torch.rand(S, 3, device="cuda") * 0.3 - 0.15).requires_grad_(True)
spf = S // 2
prec = mlp.PREC_BF16


def step():
    loss = 0
    feat = None
    if "base" in which:
        sdf, feat = mlp.run_chain(mlp.NET_FG_BASE, prec, P, x, spf, conds={0: fr["code_base"], 4: fr["code_base"]}, export_layer=8)
        loss = loss + sdf.sum()
    if "color" in which and feat is not None:
        loss = loss + mlp.run_chain(mlp.NET_FG_COLOR, prec, P, x, spf, conds={0: fr["code_color"], 3: fr["appr_code"]}, ext=feat).sum()
    if "feat" in which:
        loss = loss + mlp.run_chain(mlp.NET_FEAT, prec, P, x, spf).sum()
    if "vis" in which:
        loss = loss + mlp.run_chain(mlp.NET_VIS, prec, P, x, spf, conds={0: fr["code_vis"]}).sum()
    if "skin" in which:  # the delta-skin field (affine form unless LAB4D_SKIN_AFFINE=0)
        from lab4d_amd import warping
        raw, _ = warping.skin_logits(P, x, fr["t_articulation"], fr["t_embed"], fr["code_skin"], 2, spf, prec)
        loss = loss + raw.sum()
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
_lib.PROF = {}
n = 3
for _ in range(n):
    step()
with torch.no_grad():
    for _ in range(n):
        with _lib.timed("k_mlp_fwd<FgBase> inference", (2.0 * S * mlp.NET_MACS[0], 0.0)):
            mlp.run_chain(mlp.NET_FG_BASE, prec, P, x.detach(), spf, conds={0: fr["code_base"], 4: fr["code_base"]})
torch.cuda.synchronize()
out = {}
for k, (cnt, ms, fl, by) in sorted(_lib.prof_summary().items()):
    out[k] = {"ms": round(ms / cnt, 3), "GBps": round(by / ms / 1e6, 0) if by else None, "TFLOPs": round(fl / ms / 1e9, 0)}
print(json.dumps({"S": S, "so": os.path.basename(_lib.SO_PATH), "kernels": out}))
