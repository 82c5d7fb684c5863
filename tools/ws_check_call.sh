#!/bin/bash
# Weights-stationary chains on the GPU box (one gpurun call): bit comparison against the wave-resident kernels per net, timing of both families,
# and -- for every kernel-experiment build under gpurun_abl/ (tools/build_variants.sh =WS_TRACE ...) -- its timing / cycle trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
for net in fg_base fg_color dense dense6; do
  timeout 150 python tools/ws_compare.py --nets $net --json gpurun_out/ws_compare_$net.json 2>&1 | grep -v "^OK" | cut -c1-300 | tail -12
done
echo "######## timing"
timeout 200 python tools/ws_compare.py --nets fg_base,fg_color --quick --time 4194304 --json gpurun_out/ws_time.json 2>&1 | grep '"net"'
for v in gpurun_abl/lib_*.so; do echo "## $v"; LAB4D_WS_TRACE_PRINT=1 LAB4D_SO_PATH=$R/$v timeout 100 python tools/ws_compare.py --nets fg_base --quick --time 4194304 2>&1 | grep '"net"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:v for k,v in d.items() if k.endswith('_ms')})
    for k,v in d.get('trace_cycles_per_tile',{}).items(): print(k, v, sum(v.values()))"; done
