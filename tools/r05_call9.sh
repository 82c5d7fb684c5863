#!/bin/bash
# compile-time embedding tile in the weights-stationary backward's input-gradient math (0 spilled SGPRs): bit-equality / parity, timing, short bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mlp_ws.py tests/test_gpu_mlp.py tests/test_gpu_field.py -q 2>&1 | tail -4
timeout 200 python tools/ws_compare.py --nets fg_base,fg_color --quick --time 4194304 --json gpurun_out/r05_ws_time_dxconst.json 2>&1 | grep '"net"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['net'], {k:v for k,v in d.items() if k.endswith('_ms')})"
timeout 200 python tools/bench_chain.py 16777216 base,color 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:v['ms'] for k,v in d['kernels'].items() if '_ws' in k})"
timeout 900 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r05_bench_call9.json 2> gpurun_out/r05_bench_call9.err || tail -5 gpurun_out/r05_bench_call9.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_call9.json"))
print(d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"))
ks = d["roofline"]["kernels_ms_per_step"]
for k in ["k_mlp_fwd_ws<FgBase>", "k_mlp_bwd_ws<FgBase>", "k_mlp_fwd_ws<FgColor>", "k_mlp_bwd_ws<FgColor>", "k_mlp_bwd_ws<FgBase>@eik"]: print("  ", k, ks.get(k))
PY
