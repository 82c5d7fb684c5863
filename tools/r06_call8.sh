#!/bin/bash
# Round-6 eighth GPU call: packed-fp16 table gradient of the hash leg
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_zzhashgrid.py -q -x 2>&1 | tail -8
for mode in "" "--hash-f32-table-grad"; do
  timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --config hash $mode > gpurun_out/r06_hash${mode:+_f32grad}.json 2> gpurun_out/r06_hash${mode:+_f32grad}.err || tail -5 gpurun_out/r06_hash${mode:+_f32grad}.err
  python - "$mode" <<'PY'
import json, sys
n = "gpurun_out/r06_hash%s.json" % ("_f32grad" if sys.argv[1] else "")
try:
    d = json.load(open(n))
    print(n, d["value"], d["ms_per_step"], d["loss_last_chunk"], d["params_finite"], d["config"].get("field_rows_per_chunk"))
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:5]: print("   ", k, v)
except Exception as e:
    print(n, "FAILED", e)
PY
done
