#!/bin/bash
# HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs) over the MLP micro-bench at the launch size of the bench
# (16,777,216 samples = one 128-row chunk of a frame pair).  rocprofv3 --pmc over the whole bench.py hangs on this pool
# (observed twice, 2 x 25 GPU-minutes lost), so the counters are collected on the kernels that matter only.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD=${PMC_CMD:-"python $R/tools/bench_mlp.py 16777216 1"}   # PMC_CMD / PMC_OUT / PMC_WHAT: another micro-bench, e.g. the delta-skin chains (tools/bench_chain.py 16777216 skin)
OUT=${PMC_OUT:-r04_pmc_traffic.json}
timeout -k 5 100 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- $CMD > /tmp/pmc_fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 /tmp/pmc_fetch.log; exit 1; }
timeout -k 5 100 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- $CMD > /tmp/pmc_write.log 2>&1 || { echo "write pass failed"; tail -5 /tmp/pmc_write.log; exit 1; }
mkdir -p $R/gpurun_out
python $R/tools/pmc_summary.py /tmp/pmc_fetch /tmp/pmc_write $R/gpurun_out/$OUT "${PMC_WHAT:-rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over: tools/bench_mlp.py 16777216 1 (basefield + colourfield chains and their weight gradients at the launch size of the bench)}"
