"""Diagnostic: capture one training chunk of the bench (small shapes) and list the node types of the hipGraph (hipGraphDebugDotPrint)."""
import collections
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lab4d_amd import _lib, mlp  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4
_lib.lib()
dev = torch.device("cuda", 0)
chunks = [list(range(0, 512, 512 // rows))[:rows], list(range(1, 512, 512 // rows))[:rows]]
loop = bench.TrainLoop(dev, 512, 128, chunks, mlp.PREC_BF16, use_graph=False)
loop.st_hxy = loop.inputs[0][0].clone()
loop.st_batch = {k: v.clone() for k, v in loop.inputs[0][1].items()}
loop.st_batch["hxy"] = loop.st_hxy
loop.st_rng = bench.draw_rng(loop.M, loop.N0, loop.S0, dev, loop.gen)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    loop.chunk(loop.st_hxy, loop.st_batch, loop.st_rng)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
g.enable_debug_mode()
with torch.cuda.graph(g):
    loop.chunk(loop.st_hxy, loop.st_batch, loop.st_rng)
out = os.path.join(ROOT, "gpurun_out", "t", "chunk_graph.dot")
os.makedirs(os.path.dirname(out), exist_ok=True)
g.debug_dump(out)
txt = open(out).read()
print("dot bytes", len(txt))
labels = re.findall(r'label="([^"]*)"', txt)
kinds = collections.Counter()
for l in labels:
    k = l.split("\\n")[0][:60]
    k = re.sub(r"\d+", "#", k)
    kinds[k] += 1
for k, v in kinds.most_common(40):
    print(v, k)
ms = [l for l in labels if "emset" in l or "EMSET" in l]
print("memset nodes:", len(ms))
for l in ms[:30]:
    print("  ", l.replace("\\n", " | ")[:300])
