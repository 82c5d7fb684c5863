#!/bin/bash
# Round-5 fifth GPU call: tile-ahead prefetch in the weights-stationary chains (bit-equality, timing), the comp leg on per-chunk graphs, a short bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_mlp_ws.py -q 2>&1 | tail -4
timeout 200 python tools/ws_compare.py --nets fg_base,fg_color --quick --time 4194304 --json gpurun_out/r05_ws_time_prefetch.json 2>&1 | grep '"net"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['net'], {k:v for k,v in d.items() if k.endswith('_ms')})"
timeout 200 python tools/bench_chain.py 16777216 base,color 2>&1 | tail -4 | cut -c1-600
timeout 600 python bench.py --gpus 1 --steps 4 --warmup 2 --config comp --no-cpu-baseline --no-extras > gpurun_out/r05_comp_check.json 2> gpurun_out/r05_comp_check.err || tail -5 gpurun_out/r05_comp_check.err
timeout 900 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r05_bench_call5.json 2> gpurun_out/r05_bench_call5.err || tail -5 gpurun_out/r05_bench_call5.err
python - <<'PY'
import json
for n in ["r05_comp_check", "r05_bench_call5"]:
    try:
        d = json.load(open("gpurun_out/%s.json" % n))
        print(n, d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"), d["config"]["launch"][:30])
        ks = d["roofline"]["kernels_ms_per_step"]
        for k, v in sorted(ks.items(), key=lambda kv: -kv[1])[:6]: print("  ", k, v)
    except Exception as e:
        print(n, "FAILED", e)
PY
