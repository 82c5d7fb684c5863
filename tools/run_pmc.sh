#!/bin/bash
# HBM-traffic PMC passes over one bench step (eager launches, the bench's own chunk size). Run on the GPU box via gpurun.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline"
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- $CMD > /tmp/pmc_fetch.log 2>&1 || { tail -20 /tmp/pmc_fetch.log; exit 1; }
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- $CMD > /tmp/pmc_write.log 2>&1 || { tail -20 /tmp/pmc_write.log; exit 1; }
mkdir -p $R/gpurun_out
python $R/tools/pmc_summary.py /tmp/pmc_fetch /tmp/pmc_write $R/gpurun_out/r01_pmc_traffic.json "timeout 150 rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over: bench.py --steps 1 --warmup 0 --no-graph"
tail -2 /tmp/pmc_write.log
