#!/bin/bash
# A/B of kernel-build variants on the GPU box: bench line + per-kernel ms for each library under lab4d_amd/variants/ (and the default).
#   tools/ab_variants.sh [variant ...]      e.g. tools/ab_variants.sh il5 il4
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
run() {  # tag, so path
  LAB4D_SO_PATH=$2 timeout -s KILL 150 python bench.py --no-cpu-baseline --steps 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['roofline']['kernels_ms_per_step']
print('$1', d['value'], d['ms_per_step'], 'loss', d['loss_last_chunk'], ' '.join('%s=%.1f' % (n.replace('k_mlp_', ''), v) for n, v in sorted(k.items()) if 'bwd<' in n or 'fwd<Fg' in n))
json.dump(d, open('gpurun_out/ab_$1.json', 'w'))
"
}
run base $R/lab4d_amd/liblab4d_hip.so
for v in "$@"; do run $v $R/lab4d_amd/variants/liblab4d_hip_$v.so; done

