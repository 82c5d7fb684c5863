import sys, time, torch
sys.path.insert(0, '.')
from lab4d_amd import mlp, synthetic
P = synthetic.to_device(synthetic.make_weights(0), 'cuda')
for k in P:
    if P[k].dtype.is_floating_point: P[k].requires_grad_(True)
fr = synthetic.to_device(synthetic.add_codes(synthetic.make_frames(1, 2, 512), synthetic.make_weights(0)), 'cuda')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
x = (torch.rand(S, 3, device='cuda') * 0.3 - 0.15).requires_grad_(True)
spf = S // 2
def step():
    sdf, feat = mlp.run_chain(mlp.NET_FG_BASE, prec, P, x, spf, conds={0: fr["code_base"], 4: fr["code_base"]}, export_layer=8)
    rgb = mlp.run_chain(mlp.NET_FG_COLOR, prec, P, x, spf, conds={0: fr["code_color"], 3: fr["appr_code"]}, ext=feat)
    loss = sdf.sum() + rgb.sum()
    loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
t = time.time()
n = 3
for _ in range(n): step()
torch.cuda.synchronize()
dt = (time.time() - t) / n
flops = S * (573184 + 158464 + 37248) * 2 * 3
print(f"S={S} prec={prec} {dt*1e3:.1f} ms/step  {flops/dt/1e12:.1f} TFLOP/s  {S/128/dt/1e3:.1f} k rays/s (128 spp, base+color only)")
print(torch.cuda.max_memory_allocated()/2**30, "GiB")
