#!/bin/bash
# Round-4 A/B pass over compile-flag / small-code variants of the library (gpurun_abl/lib_<NAME>.so, built by tools/build_flagvariants.sh):
# the GPU suite on the default build first (regression gate of the round's host-side changes), then per variant the chain kernels at the
# bench's launch size (tools/bench_chain.py: HIP events per kernel) and a short whole-step bench (all kernels of the library).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
if [ "$1" != "--no-tests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r04_gpu_tests_a.txt; cat gpurun_out/r04_gpu_tests_a.txt
else shift; fi
for so in ${@:-gpurun_abl/lib_*.so}; do
  n=$(basename $so .so); n=${n#lib_}
  LAB4D_SO_PATH=$R/$so timeout 300 python tools/bench_chain.py 16777216 base,color,feat,skin,vis > gpurun_out/r04_ab_chain_$n.json 2> gpurun_out/r04_ab_chain_$n.err || tail -3 gpurun_out/r04_ab_chain_$n.err
  LAB4D_SO_PATH=$R/$so timeout 400 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r04_ab_bench_$n.json 2> gpurun_out/r04_ab_bench_$n.err || tail -3 gpurun_out/r04_ab_bench_$n.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r04_ab_chain_*.json")):
    n = os.path.basename(f)[13:-5]
    try:
        k = json.load(open(f))["kernels"]
        row = " ".join("%s=%.2f" % (a.replace("k_mlp_", "").replace("<", "_").replace(">", ""), k[a]["ms"]) for a in sorted(k) if a.startswith("k_mlp_fwd") or a.startswith("k_mlp_bwd"))
    except Exception as e:
        row = "FAILED %r" % e
    try:
        b = json.load(open("gpurun_out/r04_ab_bench_%s.json" % n))
        row += " | step %.1f ms loss %.4f finite %s" % (b["ms_per_step"], b["loss_last_chunk"], b["params_finite"])
        ks = b["roofline"]["kernels_ms_per_step"]
        row += " wgrad %.1f" % sum(v for a, v in ks.items() if "wgrad" in a)
    except Exception as e:
        row += " | bench FAILED %r" % e
    print(n, row)
PY
