cd ${GRAFT_REPO_ROOT:-/root/repo} && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_field.py tests/test_gpu_patch.py tests/test_gpu_mlp.py -q -x -k "eval or normal or eikonal or comp_eval or dispatches" 2>&1 | tail -3
timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg > gpurun_out/r04_bench_e.json 2> gpurun_out/r04_bench_e.err || tail -5 gpurun_out/r04_bench_e.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_e.json"))
e = d["eval_forward_only"]
print(d["value"], d["ms_per_step"], "eval:", e["value"], e["ms_per_call"], e["frac_of_mfma_peak"], {k: v for k, v in list(e["kernels_ms_per_call"].items())[:8]}, d["psnr_vs_ref_db"])
PY
