#!/bin/bash
# Round-4 measurement pass (one gpurun call): GPU suite, the driver's bench command, the other configurations, the RCCL world-1 leg, the per-rank share of an
# 8-rank job, rocprofv3 kernel stats of the bench command, HBM-traffic and SQ counters of the dominant kernels (separate --pmc passes).  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r04_gpu_tests.txt; cat gpurun_out/r04_gpu_tests.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err || tail -5 gpurun_out/r04_bench.err
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --config comp > gpurun_out/r04_bench_comp.json 2> gpurun_out/r04_bench_comp.err || tail -5 gpurun_out/r04_bench_comp.err
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --config multi > gpurun_out/r04_bench_multi.json 2> gpurun_out/r04_bench_multi.err || tail -5 gpurun_out/r04_bench_multi.err
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --force-dist --no-cpu-baseline --no-extras > gpurun_out/r04_bench_rccl_world1.json 2> gpurun_out/r04_bench_rccl_world1.err || tail -5 gpurun_out/r04_bench_rccl_world1.err
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --emulate-rank-of 8 --no-cpu-baseline --no-extras > gpurun_out/r04_emulate_rank_of_8.json 2> gpurun_out/r04_emulate_rank_of_8.err || tail -5 gpurun_out/r04_emulate_rank_of_8.err
python bench.py --gpus 2 > gpurun_out/r04_gpus2_on_one_gpu.txt 2>&1; echo "rc=$?" >> gpurun_out/r04_gpus2_on_one_gpu.txt
python - <<'PY'
import json
for n in ["r04_bench", "r04_bench_comp", "r04_bench_multi", "r04_bench_rccl_world1", "r04_emulate_rank_of_8"]:
    try:
        d = json.load(open("gpurun_out/%s.json" % n))
        print(n, d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"), d.get("peak_hbm_gib"), d.get("steps_discarded_by_check_grad"),
              d["roofline"]["frac"], d["roofline"].get("mfma_frac_of_peak"), d.get("whole_graph_frac_of_peak"), (d.get("cpu_baseline") or {}).get("value"),
              (d.get("fp32_leg") or {}).get("value"), (d.get("eval_forward_only") or {}).get("value"), d.get("psnr_vs_ref_db"))
    except Exception as e:
        print(n, "FAILED", e)
PY
cat gpurun_out/r04_gpus2_on_one_gpu.txt | tail -2
# rocprofv3 kernel stats of the bench command
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /tmp/prof.json 2> /tmp/prof.log || tail -20 /tmp/prof.log
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/r04_bench_kernel_stats.csv; cp /tmp/prof.json $R/gpurun_out/r04_bench_under_rocprof.json
head -6 $R/gpurun_out/r04_bench_kernel_stats.csv | cut -c1-160
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and SQ issue / wait counters of the chain kernels at the bench's launch size
cd $R && bash tools/run_pmc_mlp.sh 2>&1 | tail -4
bash tools/pmc_sq2.sh "python $R/tools/bench_chain.py 16777216 base" 2>&1 | tail -2; head -30 gpurun_out/r04_sq_counters.txt
