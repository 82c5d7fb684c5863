#!/bin/bash
# FUNCTIONAL N-rank runs of the driver's workload shape (512 x 512 x 128) on the one GPU of this box: N processes share cuda:0, the collective goes over gloo.
# Not performance numbers -- they show the N > 1 path (launcher, partition, per-rank whole-step graphs, all-reduce, timing gather) running end to end with the
# real kernels and the replicas staying bit-identical.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
for n in 2 4 8; do
  timeout 900 python bench.py --gpus $n --backend gloo --share-gpu --chunk-rows 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r06_share_gpu_$n.json 2> gpurun_out/r06_share_gpu_$n.err; echo "N=$n rc=$?"
  python - $n <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/r06_share_gpu_%s.json" % sys.argv[1]))
    print("  ", d["n_gpus"], d["rccl_ranks"], d["replicas_identical"], d["params_finite"], d["loss_last_chunk"], d["steps_discarded_by_check_grad"], d["rank_ms_per_step"], d["allreduce_ms_per_step"], d["config"]["launch"][:24])
except Exception as e:
    print("   FAILED", e)
PY
done
