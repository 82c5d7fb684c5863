#!/bin/bash
# first hardware run of the weights-stationary chains: bit comparison against the wave-resident kernels per net (one process each), then timing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
for net in fg_base fg_color dense dense6; do
  echo "######## $net"
  timeout 150 python tools/ws_compare.py --nets $net --json gpurun_out/ws_compare_$net.json 2>&1 | cut -c1-420 | tail -60
  echo "rc=$?"
done
echo "######## timing"
timeout 200 python tools/ws_compare.py --nets fg_base,fg_color --quick --time 4194304 --json gpurun_out/ws_time.json 2>&1 | tail -4
