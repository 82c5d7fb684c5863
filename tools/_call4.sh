cd ${GRAFT_REPO_ROOT:-/root/repo} && mkdir -p gpurun_out
for n in CUR ADMA1 ADMA2; do
  echo "== $n"
  LAB4D_SO_PATH=$PWD/gpurun_abl/lib_$n.so timeout 300 python -m pytest tests/test_gpu_mlp.py -q -x 2>&1 | tail -2
  LAB4D_SO_PATH=$PWD/gpurun_abl/lib_$n.so timeout 300 python tools/bench_chain.py 16777216 base,color,feat,skin,vis > gpurun_out/r04_dma_chain_$n.json 2> gpurun_out/r04_dma_chain_$n.err || tail -3 gpurun_out/r04_dma_chain_$n.err
  python - <<PY
import json
k = json.load(open("gpurun_out/r04_dma_chain_$n.json"))["kernels"]
print(" ".join("%s=%.2f" % (a.replace("k_mlp_", "").replace("<", "_").replace(">", ""), k[a]["ms"]) for a in sorted(k) if a.startswith("k_mlp_fwd") or a.startswith("k_mlp_bwd")))
PY
done
