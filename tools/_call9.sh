cd ${GRAFT_REPO_ROOT:-/root/repo} && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_zzhashgrid.py -q 2>&1 | tail -3
timeout 900 python bench.py --config hash --steps 3 --warmup 1 > gpurun_out/r04_bench_hash.json 2> gpurun_out/r04_bench_hash.err || tail -5 gpurun_out/r04_bench_hash.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_hash.json"))
print(d["value"], d["ms_per_step"], d["loss_last_chunk"], d["params_finite"], d["peak_hbm_gib"])
print({k: v for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]})
PY
