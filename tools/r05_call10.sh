#!/bin/bash
# same-box A/B: the backward's compile-time embedding tile (shipped build) against the previous commit's library (gpurun_abl/lib_PREV.so), interleaved twice
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
for rep in 1 2; do
  for so in "" gpurun_abl/lib_PREV.so; do
    n=NEW; [ -n "$so" ] && n=PREV
    if [ -n "$so" ]; then export LAB4D_SO_PATH=$R/$so; else unset LAB4D_SO_PATH; fi
    timeout 200 python tools/bench_chain.py 16777216 base,color 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$n rep$rep', {k:v['ms'] for k,v in d['kernels'].items() if 'bwd_ws' in k or 'fwd_ws' in k})"
  done
done
unset LAB4D_SO_PATH
