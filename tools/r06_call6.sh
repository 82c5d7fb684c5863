#!/bin/bash
# Round-6 sixth GPU call: the hash leg with the field on the inside-box samples only
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_zzhashgrid.py -q -x 2>&1 | tail -8
for mode in "" "--hash-no-compact"; do
  timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --config hash $mode > gpurun_out/r06_hash${mode:+_nocompact}.json 2> gpurun_out/r06_hash${mode:+_nocompact}.err || tail -5 gpurun_out/r06_hash${mode:+_nocompact}.err
  python - "$mode" <<'PY'
import json, sys
n = "gpurun_out/r06_hash%s.json" % ("_nocompact" if sys.argv[1] else "")
try:
    d = json.load(open(n))
    print(n, d["value"], d["ms_per_step"], d["loss_last_chunk"], d["params_finite"], d["config"].get("field_rows_per_chunk"), d["config"].get("inside_box_fraction"))
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]: print("   ", k, v)
    print("   roofline", d.get("roofline"))
except Exception as e:
    print(n, "FAILED", e)
PY
done
