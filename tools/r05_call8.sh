#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_mlp_ws.py -q -x 2>&1 | tail -3
timeout 300 python tools/bench_chain.py 16777216 feat,vis,skin 2>&1 | tail -2 | cut -c1-900
timeout 900 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r05_bench_call8.json 2> gpurun_out/r05_bench_call8.err || tail -5 gpurun_out/r05_bench_call8.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_call8.json"))
print(d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"))
ks = d["roofline"]["kernels_ms_per_step"]
for k in ["k_mlp_fwd<Feat>", "k_mlp_fwd<Vis>", "k_mlp_fwd<SkinA> inference", "k_mlp_bwd<Feat>", "k_mlp_bwd<Vis>", "k_mlp_fwd_ws<FgBase>", "k_mlp_bwd_ws<FgBase>"]: print("  ", k, ks.get(k))
PY
