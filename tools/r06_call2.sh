#!/bin/bash
# Round-6 second GPU call: timing and LDS counters of the default (swizzled) build next to the -DLAB4D_WS_SWZ=0 build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
echo "######## timing (ms per 4.2 M samples), three alternating passes"
for rep in 1 2 3; do
for v in default swz0; do
  if [ $v = default ]; then unset LAB4D_SO_PATH; else export LAB4D_SO_PATH=$R/gpurun_abl/lib_swz0.so; fi
  echo -n "$v: "
  timeout 200 python tools/bench_chain.py 4194304 base,color 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:v['ms'] for k,v in d['kernels'].items() if 'ws' in k or 'inference' in k})"
done
done 2>&1 | tee gpurun_out/r06_swizzle_timing.txt
echo "######## LDS counters"
cd /tmp && export TMPDIR=/tmp
for v in default swz0; do
  if [ $v = default ]; then unset LAB4D_SO_PATH; else export LAB4D_SO_PATH=$R/gpurun_abl/lib_swz0.so; fi
  rm -rf /tmp/lds_$v
  timeout -k 5 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d /tmp/lds_$v -- python $R/tools/bench_chain.py 4194304 base,color > /tmp/lds_$v.log 2>&1 || tail -3 /tmp/lds_$v.log
  python - $v <<'PY'
import csv, glob, re, sys
from collections import defaultdict
v = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob('/tmp/lds_%s/**/*counter_collection.csv' % v, recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if not re.search('k_mlp_fwd_ws|k_mlp_bwd_ws', k): continue
        k = re.sub(r'lab4d::', '', k.split('(')[0])[:60]
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(acc):
    c = {n: sum(x) / len(x) for n, x in acc[k].items()}
    print(v, k, {n: '%.4g' % x for n, x in sorted(c.items())}, 'conflict/active = %.3f' % (c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
PY
done 2>&1 | grep -v "^[EW]2026" | tee $R/gpurun_out/r06_lds_swizzle.txt
