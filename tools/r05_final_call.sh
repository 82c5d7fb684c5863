#!/bin/bash
# Round-5 measurement pass (one gpurun call): GPU suite, the driver's bench command (with its fp32 / eval / comp / multi / hash legs and the in-run PMC
# traffic), the per-rank share of an 8-rank job, `--gpus 2` on a 1-GPU box, rocprofv3 kernel stats of the bench command, HBM-traffic and SQ counters of the
# dominant kernels (separate --pmc passes).  Outputs under gpurun_out/ (copied to profiles/ by the builder).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r05_gpu_tests.txt; cat gpurun_out/r05_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err || tail -5 gpurun_out/r05_bench.err
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --config comp --no-cpu-baseline > gpurun_out/r05_bench_comp.json 2> gpurun_out/r05_bench_comp.err || tail -5 gpurun_out/r05_bench_comp.err
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --config multi --no-cpu-baseline > gpurun_out/r05_bench_multi.json 2> gpurun_out/r05_bench_multi.err || tail -5 gpurun_out/r05_bench_multi.err
timeout 600 python bench.py --gpus 1 --steps 4 --warmup 1 --config hash > gpurun_out/r05_bench_hash.json 2> gpurun_out/r05_bench_hash.err || tail -5 gpurun_out/r05_bench_hash.err
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --emulate-rank-of 8 --no-cpu-baseline --no-extras > gpurun_out/r05_emulate_rank_of_8.json 2> gpurun_out/r05_emulate_rank_of_8.err || tail -5 gpurun_out/r05_emulate_rank_of_8.err
python bench.py --gpus 2 > gpurun_out/r05_gpus2_on_one_gpu.txt 2>&1; echo "rc=$?" >> gpurun_out/r05_gpus2_on_one_gpu.txt
python - <<'PY'
import json
for n in ["r05_bench", "r05_bench_comp", "r05_bench_multi", "r05_emulate_rank_of_8"]:
    try:
        d = json.load(open("gpurun_out/%s.json" % n))
        print(n, d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"), d.get("peak_hbm_gib"), d.get("steps_discarded_by_check_grad"),
              d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("mfma_frac_of_peak"), d["roofline"].get("traffic"), d.get("whole_graph_frac_of_peak"),
              (d.get("cpu_baseline") or {}).get("value"), (d.get("fp32_leg") or {}).get("value"), (d.get("eval_forward_only") or {}).get("value"), d.get("psnr_vs_ref_db"),
              {k: v.get("value") for k, v in (d.get("other_configs") or {}).items()})
    except Exception as e:
        print(n, "FAILED", e)
try:
    d = json.load(open("gpurun_out/r05_bench_hash.json")); print("hash", d["value"], d["ms_per_step"], d["config"].get("inside_box_fraction"))
except Exception as e:
    print("hash FAILED", e)
PY
tail -2 gpurun_out/r05_gpus2_on_one_gpu.txt
# rocprofv3 kernel stats of the bench command
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /tmp/prof.json 2> /tmp/prof.log || tail -20 /tmp/prof.log
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/r05_bench_kernel_stats.csv; cp /tmp/prof.json $R/gpurun_out/r05_bench_under_rocprof.json
head -8 $R/gpurun_out/r05_bench_kernel_stats.csv | cut -c1-160
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and SQ issue / wait counters of the chain kernels at the bench's launch size
cd $R && PMC_OUT=r05_pmc_traffic.json bash tools/run_pmc_mlp.sh 2>&1 | tail -4
SQ_SKIP_LIST=1 SQ_OUT=r05_sq_counters.txt bash tools/pmc_sq2.sh "python $R/tools/bench_chain.py 16777216 base" 2>&1 | tail -2; head -34 gpurun_out/r05_sq_counters.txt
timeout 120 python tools/clock_under_load.py 4194304 4 > gpurun_out/r05_clock_under_load.json 2>/dev/null; cat gpurun_out/r05_clock_under_load.json | cut -c1-200
timeout 300 python tools/count_step_launches.py 2>/dev/null | tail -1 | tee gpurun_out/r05_step_launches.json
