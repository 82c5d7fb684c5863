#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/ws_bench_on.json 2> gpurun_out/ws_bench_on.err || tail -5 gpurun_out/ws_bench_on.err
LAB4D_WS=0 timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/ws_bench_off.json 2> gpurun_out/ws_bench_off.err || tail -5 gpurun_out/ws_bench_off.err
python - <<'PY'
import json
for n in ["ws_bench_on", "ws_bench_off"]:
    try:
        d = json.load(open("gpurun_out/%s.json" % n))
        print(n, d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"), d.get("peak_hbm_gib"), d["roofline"]["kernel"], d["roofline"]["avg_ms"], d["roofline"].get("mfma_frac_of_peak"))
        ks = d["roofline"].get("kernels_ms_per_step", {})
        print({k: v for k, v in ks.items() if "Fg" in k})
    except Exception as e:
        print(n, "FAILED", e)
PY
