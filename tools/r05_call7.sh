#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
LAB4D_PARITY_RECORD=new timeout 600 python -m pytest tests/test_gpu_field.py -q -rf -k "comp_eval" 2>&1 | tail -12
timeout 300 python - <<'PY'
import json, torch, bench
from lab4d_amd import _lib
_lib.lib()
print(json.dumps(bench.psnr_vs_reference(torch.device("cuda", 0))))
PY
