#!/bin/bash
# Round-4 parity pass (one gpurun call): record the MI355X measurements of the NEW parity entries (full-size C3 / C4 fixtures, the comp tests moved to
# the check() regime), merge them into tests/golden/parity_measured.json on the box, then run the whole GPU suite with the assertions on
# (bounds from the merged table + the same-tensor fp32 noise floors), then the default bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
LAB4D_PARITY_RECORD=1 timeout 600 python -m pytest tests/test_gpu_field.py -q -k "multi10_bench or comp_training_graph_at_the_bench_shape_fp32 or comp_train_matches or unshared" 2>&1 | tail -5
python - <<'PY'
import json
new = json.load(open("gpurun_out/parity_measured.json"))
old = json.load(open("tests/golden/parity_measured.json"))
old.update(new)
json.dump(old, open("tests/golden/parity_measured.json", "w"), indent=1, sort_keys=True)
json.dump(old, open("gpurun_out/parity_measured_merged.json", "w"), indent=1, sort_keys=True)
print("recorded tags:", sorted(new))
PY
timeout 1200 python -m pytest tests -m gpu -q -rf 2>&1 | tail -40 > gpurun_out/r04_gpu_tests_b.txt; tail -25 gpurun_out/r04_gpu_tests_b.txt
timeout 900 python bench.py --gpus 1 --steps 6 --warmup 2 > gpurun_out/r04_bench_b.json 2> gpurun_out/r04_bench_b.err || tail -5 gpurun_out/r04_bench_b.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_b.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "loss_last_chunk", "params_finite", "psnr_vs_ref_db", "fp32_leg")})
print("eval:", {k: v for k, v in (d.get("eval_forward_only") or {}).items() if k != "what"})
PY
