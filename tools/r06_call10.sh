#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_zzhashgrid.py -q 2>&1 | tail -3
for lm in 1 0; do
  LAB4D_HASH_LEVEL_MAJOR=$lm timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --config hash > gpurun_out/r06_hash_lm$lm.json 2> gpurun_out/r06_hash_lm$lm.err || tail -5 gpurun_out/r06_hash_lm$lm.err
  python - $lm <<'PY'
import json, sys
n = "gpurun_out/r06_hash_lm%s.json" % sys.argv[1]
try:
    d = json.load(open(n))
    print(n, d["value"], d["ms_per_step"], d["loss_last_chunk"], d["params_finite"])
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:4]: print("   ", k, v)
except Exception as e:
    print(n, "FAILED", e)
PY
done
