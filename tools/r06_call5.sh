#!/bin/bash
# Round-6 fifth GPU call: what a byte-eliminating design could buy at most, MEASURED -- the weight-gradient kernel with its operands served by the L2
# (-DLAB4D_ABL_WGRAD_L2: timing only) and the chain kernels with their activation / dZ stores removed (-DLAB4D_WSABL_NOFLUSH: timing only), next to the
# shipped build at the bench's launch size; clocks / power under each.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
export LAB4D_ALLOW_EXPERIMENT_BUILD=1
for rep in 1 2; do
for v in default wl2 noflush; do
  if [ $v = default ]; then unset LAB4D_SO_PATH; else export LAB4D_SO_PATH=$R/gpurun_abl/lib_$v.so; fi
  echo -n "$v: "
  timeout 300 python tools/bench_chain.py 16777216 base,color 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(json.dumps({k:v['ms'] for k,v in d['kernels'].items() if 'ws' in k or 'wgrad_dma<8' in k or 'inference' in k}))"
done
done 2>&1 | tee gpurun_out/r06_byte_elimination_ceiling.txt
