cd ${GRAFT_REPO_ROOT:-/root/repo} && mkdir -p gpurun_out
LAB4D_PARITY_RECORD=1 timeout 900 python -m pytest tests/test_gpu_field.py tests/test_gpu_zoptim.py -q 2>&1 | tail -3
python - <<'PY'
import json
new = json.load(open("gpurun_out/parity_measured.json")); old = json.load(open("tests/golden/parity_measured.json")); old.update(new)
json.dump(old, open("tests/golden/parity_measured.json", "w"), indent=1, sort_keys=True); json.dump(old, open("gpurun_out/parity_measured_merged.json", "w"), indent=1, sort_keys=True)
print("recorded tags:", len(new))
PY
timeout 1200 python -m pytest tests -m gpu -q -rf 2>&1 | tail -30 | cut -c1-2500 > gpurun_out/r04_gpu_tests_c.txt; tail -12 gpurun_out/r04_gpu_tests_c.txt
