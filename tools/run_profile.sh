#!/bin/bash
# Kernel-trace profile of the default bench command + a plain bench run. Run on the GPU box via gpurun.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> /tmp/bench.err || tail -20 /tmp/bench.err
cat $R/gpurun_out/${TAG}_bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu-baseline > /tmp/prof.log 2>&1 || tail -20 /tmp/prof.log
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/${TAG}_bench_kernel_stats.csv
head -25 $R/gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-160
