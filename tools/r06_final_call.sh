#!/bin/bash
# Round-6 measurement pass (one gpurun call): GPU suite, smoke, the driver's bench command (with its sustained / fp32 / eval / comp / multi / hash legs, the
# in-run PMC traffic and the 8,192-ray CPU baseline), the per-rank share of an 8-rank job, `--gpus 2` on a 1-GPU box, comp under the RCCL process group,
# rocprofv3 kernel stats of the bench command, HBM-traffic and SQ counters of the dominant kernels (separate --pmc passes).
# Outputs under gpurun_out/ (copied to profiles/ by the builder).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r06_gpu_tests.txt; cat gpurun_out/r06_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$(date +%s)
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err || tail -5 gpurun_out/r06_bench.err
echo "bench wall: $(( $(date +%s) - t0 )) s"
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --emulate-rank-of 8 --no-cpu-baseline --no-extras > gpurun_out/r06_emulate_rank_of_8.json 2> gpurun_out/r06_emulate_rank_of_8.err || tail -5 gpurun_out/r06_emulate_rank_of_8.err
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --force-dist --no-cpu-baseline --no-extras > gpurun_out/r06_bench_rccl_world1.json 2> gpurun_out/r06_bench_rccl_world1.err || tail -5 gpurun_out/r06_bench_rccl_world1.err
timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --config comp --force-dist --no-cpu-baseline --no-extras > gpurun_out/r06_comp_rccl_world1.json 2> gpurun_out/r06_comp_rccl_world1.err || tail -5 gpurun_out/r06_comp_rccl_world1.err
python bench.py --gpus 2 > gpurun_out/r06_gpus2_on_one_gpu.txt 2>&1; echo "rc=$?" >> gpurun_out/r06_gpus2_on_one_gpu.txt
python - <<'PY'
import json
for n in ["r06_bench", "r06_emulate_rank_of_8", "r06_bench_rccl_world1", "r06_comp_rccl_world1"]:
    try:
        d = json.load(open("gpurun_out/%s.json" % n))
        print(n, d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"), d.get("peak_hbm_gib"), d.get("steps_discarded_by_check_grad"),
              d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("mfma_frac_of_peak"), d["roofline"].get("traffic"), d.get("whole_graph_frac_of_peak"),
              (d.get("sustained") or {}).get("value"), (d.get("cpu_baseline") or {}).get("value"), (d.get("fp32_leg") or {}).get("value"),
              (d.get("eval_forward_only") or {}).get("value"), d.get("psnr_vs_ref_db"), {k: v.get("value") for k, v in (d.get("other_configs") or {}).items()})
    except Exception as e:
        print(n, "FAILED", e)
PY
tail -2 gpurun_out/r06_gpus2_on_one_gpu.txt
# rocprofv3 kernel stats of the bench command
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /tmp/prof.json 2> /tmp/prof.log || tail -20 /tmp/prof.log
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/r06_bench_kernel_stats.csv; cp /tmp/prof.json $R/gpurun_out/r06_bench_under_rocprof.json
head -8 $R/gpurun_out/r06_bench_kernel_stats.csv | cut -c1-160
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and SQ issue / wait counters of the chain kernels at the bench's launch size
cd $R && PMC_OUT=r06_pmc_traffic.json bash tools/run_pmc_mlp.sh 2>&1 | tail -4
SQ_SKIP_LIST=1 SQ_OUT=r06_sq_counters.txt bash tools/pmc_sq2.sh "python $R/tools/bench_chain.py 16777216 base,color" 2>&1 | tail -2; head -34 gpurun_out/r06_sq_counters.txt
timeout 120 python tools/clock_under_load.py 4194304 4 > gpurun_out/r06_clock_under_load.json 2>/dev/null; cat gpurun_out/r06_clock_under_load.json | cut -c1-200
