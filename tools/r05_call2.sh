#!/bin/bash
# Round-5 second GPU call: weights-stationary chains with the fixed block counters + the per-SIMD matrix-pipe token (bit-equality first, then the
# 2 x 2 timing matrix sync x token and the cycle trace), the GPU suite (first-run recording of the new parity tags), the ring probe's cache-scope modes,
# a short bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
echo "######## ws bit-equality"
timeout 600 python -m pytest tests/test_gpu_mlp_ws.py -q 2>&1 | tail -8 | tee gpurun_out/r05_ws_tests.txt
echo "######## ws timing: shipped (counters + token)"
timeout 200 python tools/ws_compare.py --nets fg_base,fg_color --quick --time 4194304 --json gpurun_out/r05_ws_time_shipped.json 2>&1 | grep '"net"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['net'], {k:v for k,v in d.items() if k.endswith('_ms')})"
export LAB4D_ALLOW_EXPERIMENT_BUILD=1
for v in gpurun_abl/lib_*.so; do echo "## $v"; LAB4D_WS_TRACE_PRINT=1 LAB4D_SO_PATH=$R/$v timeout 150 python tools/ws_compare.py --nets fg_base,fg_color --quick --time 4194304 --json gpurun_out/r05_ws_time_$(basename $v .so).json 2>&1 | grep '"net"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['net'], {k:v for k,v in d.items() if k.endswith('_ms')})
    if 'TRACE' in '$v':
        for k,v in d.get('trace_cycles_per_tile',{}).items(): print(k, v, sum(v.values()))"; done
unset LAB4D_ALLOW_EXPERIMENT_BUILD
echo "######## GPU suite (new parity tags recorded)"
LAB4D_PARITY_RECORD=new timeout 1200 python -m pytest tests -m gpu -q -rf --deselect tests/test_gpu_mlp_ws.py 2>&1 | tail -40 | tee gpurun_out/r05_gpu_tests_call2.txt
echo "######## ring probe"
timeout 400 tools/probes/ring_probe.bin 1024 | tee gpurun_out/r05_ring_probe.jsonl
echo "######## short bench"
timeout 900 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r05_bench_call2.json 2> gpurun_out/r05_bench_call2.err || tail -5 gpurun_out/r05_bench_call2.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_call2.json"))
print(d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"))
ks = d["roofline"]["kernels_ms_per_step"]
for k, v in sorted(ks.items(), key=lambda kv: -kv[1])[:12]: print("  ", k, v)
PY
