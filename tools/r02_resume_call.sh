#!/bin/bash
# One gpurun call: full GPU suite, bench line, rocprofv3 kernel stats of the same command, launch count.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r02_gpu_tests.txt
tail -3 gpurun_out/r02_gpu_tests.txt
timeout 300 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err || tail -20 gpurun_out/r02_bench.err
tail -c 3000 gpurun_out/r02_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu-baseline > /tmp/prof.log 2>&1 || tail -20 /tmp/prof.log
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/r02_bench_kernel_stats.csv
head -40 $R/gpurun_out/r02_bench_kernel_stats.csv | cut -c1-170
cd $R && timeout 120 python tools/count_launches.py > gpurun_out/r02_launches.txt 2>&1; tail -30 gpurun_out/r02_launches.txt
