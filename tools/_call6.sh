cd ${GRAFT_REPO_ROOT:-/root/repo} && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_skinning.py -q -x -rf 2>&1 | tail -25 | cut -c1-1800
for f in 1 0; do
  LAB4D_FUSED_NARROW=$f timeout 300 python tools/bench_chain.py 16777216 skin,vis > gpurun_out/r04_fused_narrow_$f.json 2> gpurun_out/r04_fused_narrow_$f.err || tail -5 gpurun_out/r04_fused_narrow_$f.err
  python - <<PY
import json
k = json.load(open("gpurun_out/r04_fused_narrow_$f.json"))["kernels"]
print("fused=$f", " ".join("%s=%.2f" % (a, k[a]["ms"]) for a in sorted(k)))
PY
done
