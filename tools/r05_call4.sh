#!/bin/bash
# Round-5 fourth GPU call: the whole-step hipGraph pair (tests, then A/B against the per-chunk graph on the bench command, the 8-rank share and the
# RCCL world-1 leg), the eval-bench parity tests with the opacity-weighted L2 normal metric (first-run recording).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
echo "######## loop tests"
timeout 900 python -m pytest tests/test_gpu_zzbench_loop.py tests/test_gpu_ztrajectory.py -q -rf 2>&1 | tail -15
echo "######## eval-bench parity"
LAB4D_PARITY_RECORD=new timeout 600 python -m pytest tests/test_gpu_field.py -q -rf -k "eval_graph_at_the_bench_size" 2>&1 | tail -12
echo "######## bench A/B"
for mode in "" "--chunk-graph"; do
  n=step; [ -n "$mode" ] && n=chunk
  timeout 600 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-extras $mode > gpurun_out/r05_ab_${n}graph.json 2> gpurun_out/r05_ab_${n}graph.err || tail -5 gpurun_out/r05_ab_${n}graph.err
  timeout 600 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-extras --emulate-rank-of 8 $mode > gpurun_out/r05_ab_${n}graph_rank8.json 2> gpurun_out/r05_ab_${n}graph_rank8.err || tail -5 gpurun_out/r05_ab_${n}graph_rank8.err
done
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --force-dist --no-cpu-baseline --no-extras > gpurun_out/r05_bench_rccl_world1.json 2> gpurun_out/r05_bench_rccl_world1.err || tail -5 gpurun_out/r05_bench_rccl_world1.err
python - <<'PY'
import json
for n in ["r05_ab_stepgraph", "r05_ab_chunkgraph", "r05_ab_stepgraph_rank8", "r05_ab_chunkgraph_rank8", "r05_bench_rccl_world1"]:
    try:
        d = json.load(open("gpurun_out/%s.json" % n))
        print(n, d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"), d.get("peak_hbm_gib"), d.get("steps_discarded_by_check_grad"), d["config"]["launch"][:40])
    except Exception as e:
        print(n, "FAILED", e)
PY
