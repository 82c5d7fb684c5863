#!/bin/bash
# Round-6 third GPU call: the rowmlp programs (per-frame MLPs as one launch each way) against the torch algebra and the reference's pose fixture
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rowmlp.py tests/test_gpu_zpose.py tests/test_gpu_patch.py -q -x 2>&1 | tail -25
