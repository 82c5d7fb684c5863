#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python bench.py --gpus 1 --steps 4 --warmup 2 --force-dist --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('force-dist', d['value'], d['params_finite'], d['rccl_ranks'], d['replicas_identical'], d['allreduce_ms_per_step'])"
timeout 900 python -m pytest tests/test_gpu_zzbench_loop.py -q 2>&1 | tail -2
