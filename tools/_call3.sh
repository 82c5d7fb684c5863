cd ${GRAFT_REPO_ROOT:-/root/repo} && mkdir -p gpurun_out
LAB4D_PARITY_RECORD=1 timeout 600 python -m pytest tests/test_gpu_field.py -q -k "bench_shape_multi10" 2>&1 | tail -2
python - <<'PY'
import json
new = json.load(open("gpurun_out/parity_measured.json")); old = json.load(open("tests/golden/parity_measured.json")); old.update(new)
json.dump(old, open("tests/golden/parity_measured.json", "w"), indent=1, sort_keys=True); json.dump(old, open("gpurun_out/parity_measured_merged.json", "w"), indent=1, sort_keys=True)
print("recorded tags:", sorted(new))
PY
timeout 900 python -m pytest tests/test_gpu_field.py tests/test_gpu_mlp.py tests/test_gpu_patch.py -q -rf 2>&1 | tail -30 | cut -c1-3000
timeout 900 python tools/diag/grad_terms.py comp_bench 2>&1 | tail -70
