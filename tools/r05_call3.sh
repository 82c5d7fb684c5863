#!/bin/bash
# Round-5 third GPU call: the GPU suite on the shipped (barrier) build with first-run recording of the new parity tags, the ring probe with its waits
# tied to the loaded registers, the skinning micro-bench + SQ counters of the blend kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
echo "######## GPU suite (new parity tags recorded)"
LAB4D_PARITY_RECORD=new timeout 1200 python -m pytest tests -m gpu -q -rf 2>&1 | tail -40 | tee gpurun_out/r05_gpu_tests_call3.txt
echo "######## ring probe"
timeout 400 tools/probes/ring_probe.bin 1024 | tee gpurun_out/r05_ring_probe.jsonl | grep -v '"mode": 0'
echo "######## skinning micro-bench"
timeout 300 python tools/bench_skin.py 16777216 3 | tee gpurun_out/r05_bench_skin.json
SQ_FILTER='k_blend|k_mlp_bwd_fused|k_mlp_fwd' SQ_OUT=r05_sq_skin.txt timeout 900 bash tools/pmc_sq2.sh "python $R/tools/bench_skin.py 16777216 2" 2>&1 | tail -3
head -120 gpurun_out/r05_sq_skin.txt
