#!/bin/bash
# Round-5 first GPU call: (1) the weights-stationary chains with per-block progress counters -- bit-equality incl. the multi-tile cases FIRST (a sync bug
# must show up as a failed comparison or a trap, under a short timeout), then timing against the barrier build and the per-wave cycle traces of both;
# (2) the whole GPU suite with first-run recording of the new parity tags; (3) the CU -> CU ring probe; (4) the driver's bench command.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
echo "######## ws bit-equality"
timeout 400 python -m pytest tests/test_gpu_mlp_ws.py -x -q 2>&1 | tail -6 | tee gpurun_out/r05_ws_tests.txt
echo "######## ws timing: counters (shipped) vs barrier"
timeout 200 python tools/ws_compare.py --nets fg_base,fg_color --quick --time 4194304 --json gpurun_out/r05_ws_time_sync1.json 2>&1 | grep '"net"'
export LAB4D_ALLOW_EXPERIMENT_BUILD=1
for v in gpurun_abl/lib_*.so; do echo "## $v"; LAB4D_WS_TRACE_PRINT=1 LAB4D_SO_PATH=$R/$v timeout 150 python tools/ws_compare.py --nets fg_base,fg_color --quick --time 4194304 --json gpurun_out/r05_ws_time_$(basename $v .so).json 2>&1 | grep '"net"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['net'], {k:v for k,v in d.items() if k.endswith('_ms')})
    for k,v in d.get('trace_cycles_per_tile',{}).items(): print(k, v, sum(v.values()))"; done
unset LAB4D_ALLOW_EXPERIMENT_BUILD
echo "######## GPU suite (new parity tags recorded)"
LAB4D_PARITY_RECORD=new timeout 1200 python -m pytest tests -m gpu -q -rf --deselect tests/test_gpu_mlp_ws.py 2>&1 | tail -40 | tee gpurun_out/r05_gpu_tests_call1.txt
echo "######## ring probe"
timeout 300 tools/probes/ring_probe.bin 1024 | tee gpurun_out/r05_ring_probe.jsonl
echo "######## bench (driver command)"
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_call1.json 2> gpurun_out/r05_bench_call1.err || tail -5 gpurun_out/r05_bench_call1.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_call1.json"))
print(d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"), d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("traffic_source"))
print("fp32", (d.get("fp32_leg") or {}).get("value"), "eval", (d.get("eval_forward_only") or {}).get("value"))
print("others", {k: (v.get("value"), v.get("wall_s"), v.get("error")) for k, v in (d.get("other_configs") or {}).items()})
ks = d["roofline"]["kernels_ms_per_step"]
for k, v in sorted(ks.items(), key=lambda kv: -kv[1])[:14]: print("  ", k, v)
PY
