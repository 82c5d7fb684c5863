cd ${GRAFT_REPO_ROOT:-/root/repo} && mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
LAB4D_FUSED_VIS=1 timeout 300 python -m pytest tests/test_gpu_mlp.py -q -k "fused_narrow or chain_forward_backward" 2>&1 | tail -2
for f in 1 0; do
LAB4D_FUSED_NARROW=$f timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r04_bench_fused$f.json 2> gpurun_out/r04_bench_fused$f.err || tail -5 gpurun_out/r04_bench_fused$f.err
python - <<PY
import json
d = json.load(open("gpurun_out/r04_bench_fused$f.json")); ks = d["roofline"]["kernels_ms_per_step"]
print("fused=$f", d["value"], d["ms_per_step"], d["loss_last_chunk"], d["peak_hbm_gib"], {k: v for k, v in ks.items() if "Skin" in k or "<2,1>" in k or "<1," in k})
PY
done
