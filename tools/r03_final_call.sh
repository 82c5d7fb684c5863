#!/bin/bash
# Round-3 measurement pass (one gpurun call): GPU test suite, the driver's bench command, the other configurations, the RCCL world-1 leg,
# rocprofv3 kernel stats of the same workload.  Outputs under gpurun_out/ (copied into profiles/ by hand).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r03_gpu_tests.txt; cat gpurun_out/r03_gpu_tests.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || tail -5 gpurun_out/r03_bench.err
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --config comp > gpurun_out/r03_bench_comp.json 2> gpurun_out/r03_bench_comp.err || tail -5 gpurun_out/r03_bench_comp.err
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --config multi > gpurun_out/r03_bench_multi.json 2> gpurun_out/r03_bench_multi.err || tail -5 gpurun_out/r03_bench_multi.err
timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --dtype f32 --no-cpu-baseline > gpurun_out/r03_bench_f32.json 2> gpurun_out/r03_bench_f32.err || tail -5 gpurun_out/r03_bench_f32.err
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --force-dist --no-cpu-baseline > gpurun_out/r03_bench_rccl_world1.json 2> gpurun_out/r03_bench_rccl_world1.err || tail -5 gpurun_out/r03_bench_rccl_world1.err
python - <<'PY'
import json
for n in ["r03_bench", "r03_bench_comp", "r03_bench_multi", "r03_bench_f32", "r03_bench_rccl_world1"]:
    try:
        d = json.load(open("gpurun_out/%s.json" % n))
        print(n, d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"), d.get("peak_hbm_gib"), d.get("steps_discarded_by_check_grad"),
              d["roofline"]["frac"], d["roofline"].get("mfma_frac_of_peak"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "FAILED", e)
PY
bash tools/r03_profile_call.sh 3
