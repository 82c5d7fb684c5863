#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_check.json 2> gpurun_out/r06_bench_check.err || tail -5 gpurun_out/r06_bench_check.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_check.json"))
print(d["value"], d["ms_per_step"], (d.get("sustained") or {}).get("value"))
for k, v in d["other_configs"].items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("steps"), v.get("wall_s"), v.get("error"), (v.get("workload") or "")[:60])
PY
