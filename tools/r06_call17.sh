#!/bin/bash
# BASELINE configs[0]'s shape (64 x 64 crop of a frame pair, 64 samples/ray = 8,192 rays x 64 per step: the reference's own CPU-runnable case, and close to what the
# reference's trainer feeds per step) through the same training loop: whole-step graphs vs eager launches.  A launch-bound regime: what the graphs are for.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
for mode in "" "--no-graph"; do
  timeout 600 python bench.py --res 64 --spp 64 --chunk-rows 64 --steps 50 --warmup 5 --no-extras --no-cpu-baseline $mode > gpurun_out/r06_config0_shape${mode:+_eager}.json 2> gpurun_out/r06_config0_shape${mode:+_eager}.err
  python - "$mode" <<'PY'
import json, sys
n = "gpurun_out/r06_config0_shape%s.json" % ("_eager" if sys.argv[1] else "")
try:
    d = json.load(open(n)); print(n, d["value"], d["ms_per_step"], d["config"]["rays_per_step"], d["config"]["launch"][:34], d["loss_last_chunk"], d["params_finite"])
except Exception as e:
    print(n, "FAILED", e)
PY
done
