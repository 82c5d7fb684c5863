#!/bin/bash
# Round-6 call 12: the 128-wide nets (feature field, background basefield) on the weights-stationary family: bit-equality and timing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mlp_ws.py -q 2>&1 | tail -12
echo "######## timing feat (ms per 16.7 M samples), WS on / off"
for ws in 1 0; do
  echo -n "LAB4D_WS=$ws: "
  LAB4D_WS=$ws timeout 300 python tools/bench_chain.py 16777216 feat 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(json.dumps({k:v['ms'] for k,v in d['kernels'].items() if 'Feat' in k}))"
done
