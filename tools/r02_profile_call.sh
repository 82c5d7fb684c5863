#!/bin/bash
# bench line + rocprofv3 kernel stats of the same command (run on the GPU box via gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -s KILL 300 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err || tail -20 gpurun_out/r02_bench.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu-baseline > /tmp/prof.log 2>&1 || tail -20 /tmp/prof.log
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/r02_bench_kernel_stats.csv
head -5 $R/gpurun_out/r02_bench_kernel_stats.csv | cut -c1-150
