#!/bin/bash
# First gpurun call of the next round, in one go: the full GPU suite with the opt-in tests enabled, the timing of the rows that
# round 1 left untimed, and a fresh bench line + kernel stats.  Everything lands in gpurun_out/.
#   /usr/local/graft/bin/gpurun --timeout 900 -- tools/r02_first_call.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
LAB4D_RUN_UNVALIDATED=1 timeout 400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r02_gpu_tests.txt
timeout 120 python tools/bench_widened.py > gpurun_out/r02_widened_timing.jsonl 2>&1
timeout 300 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
tail -3 gpurun_out/r02_gpu_tests.txt; cat gpurun_out/r02_widened_timing.jsonl; tail -c 600 gpurun_out/r02_bench.json
