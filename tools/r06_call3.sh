#!/bin/bash
# Round-6 third GPU call: the rowmlp programs (per-frame MLPs as one launch each way) against the torch algebra and the reference's pose fixture;
# the comp configuration on the whole-step graph (explicit index_add_ backward of the bg prologue)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rowmlp.py tests/test_gpu_zpose.py tests/test_gpu_patch.py -q 2>&1 | tail -25
echo "######## comp on the whole-step graph"
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --config comp --no-cpu-baseline --no-extras > gpurun_out/r06_comp_stepgraph.json 2> gpurun_out/r06_comp_stepgraph.err; echo "rc=$?"; tail -5 gpurun_out/r06_comp_stepgraph.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r06_comp_stepgraph.json"))
    print("comp", d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"), d["config"]["launch"][:40], d.get("steps_discarded_by_check_grad"))
except Exception as e:
    print("comp FAILED", e)
PY
timeout 600 python -m pytest tests/test_gpu_zzbench_loop.py tests/test_gpu_prologue.py -q 2>&1 | tail -4
