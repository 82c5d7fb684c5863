#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rowmlp.py tests/test_gpu_zpose.py tests/test_gpu_patch.py -q 2>&1 | tail -12
timeout 300 python tools/bench_rowmlp.py 256 256 2>/dev/null | tail -1 | tee gpurun_out/r06_rowmlp_vs_torch.json
