import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lab4d_amd import mlp, synthetic
P = synthetic.to_device(synthetic.make_weights(0), 'cuda')
fr = synthetic.to_device(synthetic.add_codes(synthetic.make_frames(1, 2, 512), synthetic.make_weights(0)), 'cuda')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
x = (torch.rand(S, 3, device='cuda') * 0.3 - 0.15)
spf = S // 2
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
for prec in (1, 0):
    with torch.no_grad():
        ms = t(lambda: mlp.run_chain(mlp.NET_FG_BASE, prec, P, x, spf, conds={0: fr["code_base"], 4: fr["code_base"]}))
        print(f"prec={prec} base fwd (no stores) {ms:.2f} ms  {S*573184*2/ms/1e9:.1f} TFLOP/s")
        ms = t(lambda: mlp.run_chain(mlp.NET_VIS, prec, P, x, spf, conds={0: fr["code_vis"]}))
        print(f"prec={prec} vis fwd (no stores) {ms:.2f} ms  {S*10240*2/ms/1e9:.1f} TFLOP/s")
    xg = x.clone().requires_grad_(True)
    ms = t(lambda: mlp.run_chain(mlp.NET_FG_BASE, prec, P, xg, spf, conds={0: fr["code_base"], 4: fr["code_base"]}))
    print(f"prec={prec} base fwd (with act stores) {ms:.2f} ms")
