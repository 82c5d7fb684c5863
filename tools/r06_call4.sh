#!/bin/bash
# Round-6 fourth GPU call: host facts, rowmlp tests, the driver's bench command with the budgeted extras (sustained leg, 8,192-ray CPU baseline, 5-step
# legs), comp under the RCCL process group at world 1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
echo "host: $(nproc) threads; $(free -g | sed -n 2p)"
timeout 600 python -m pytest tests/test_gpu_rowmlp.py -q 2>&1 | tail -3
t0=$(date +%s)
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_call4.json 2> gpurun_out/r06_bench_call4.err || tail -5 gpurun_out/r06_bench_call4.err
echo "bench wall: $(( $(date +%s) - t0 )) s"
grep "bench headline" gpurun_out/r06_bench_call4.err | cut -c1-300
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r06_bench_call4.json"))
    print("bench", d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"), "sustained", d.get("sustained"))
    print("roofline", {k: d["roofline"][k] for k in ("kernel", "bound", "frac", "traffic", "traffic_source")})
    print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("sample"))
    print("fp32", (d.get("fp32_leg") or {}).get("value"), "eval", (d.get("eval_forward_only") or {}).get("value"), "psnr", d.get("psnr_vs_ref_db"))
    for k, v in (d.get("other_configs") or {}).items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("steps"), v.get("wall_s"), v.get("error"))
except Exception as e:
    print("bench FAILED", e)
PY
echo "######## comp under the RCCL process group (world 1)"
timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --config comp --force-dist --no-cpu-baseline --no-extras > gpurun_out/r06_comp_rccl_world1.json 2> gpurun_out/r06_comp_rccl_world1.err; echo "rc=$?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r06_comp_rccl_world1.json"))
    print("comp rccl", d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"), d["config"]["launch"][:40], d.get("rccl_ranks"), d.get("allreduce_ms_per_step"), d.get("rank_ms_per_step"))
except Exception as e:
    print("comp rccl FAILED", e)
PY
