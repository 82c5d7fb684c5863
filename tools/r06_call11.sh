#!/bin/bash
# the chain kernels with their transposing-read tile stores removed (-DLAB4D_WSABL_NOST: forward activations AND backward dZ; timing only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
export LAB4D_ALLOW_EXPERIMENT_BUILD=1
for rep in 1 2; do
for v in default nost; do
  if [ $v = default ]; then unset LAB4D_SO_PATH; else export LAB4D_SO_PATH=$R/gpurun_abl/lib_$v.so; fi
  echo -n "$v: "
  timeout 300 python tools/bench_chain.py 16777216 base,color 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(json.dumps({k:v['ms'] for k,v in d['kernels'].items() if 'ws' in k}))"
done
done 2>&1 | tee gpurun_out/r06_chain_no_stores.txt
