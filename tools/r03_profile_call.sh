#!/bin/bash
# rocprofv3 kernel stats of the bench command (run on the GPU box via gpurun): gpurun_out/r03_bench_kernel_stats.csv (+ the kernel trace of one run,
# reduced to per-kernel rows, for launch counting)
R=${GRAFT_REPO_ROOT:-/root/repo}
STEPS=${1:-3}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline > /tmp/prof.json 2> /tmp/prof.log || tail -20 /tmp/prof.log
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/r03_bench_kernel_stats.csv
cp /tmp/prof.json $R/gpurun_out/r03_bench_under_rocprof.json
head -4 $R/gpurun_out/r03_bench_kernel_stats.csv | cut -c1-150
