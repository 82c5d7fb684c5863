#!/bin/bash
# Round-6 first GPU call: the slot swizzle of the weights-stationary slabs (bit-equality, timing and LDS bank-conflict counters against the
# -DLAB4D_WS_SWZ=0 build under gpurun_abl/), the no-update warm-up of the whole-step graph, the whole GPU suite and a short bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
echo "######## ws bit-equality + bench loop"
timeout 900 python -m pytest tests/test_gpu_mlp_ws.py tests/test_gpu_zzbench_loop.py -q -x 2>&1 | tail -5
echo "######## timing (ms per 4.2 M samples)"
for v in "" gpurun_abl/lib_swz0.so; do
  echo "## ${v:-default}"
  LAB4D_SO_PATH=${v:+$R/$v} timeout 200 python tools/bench_chain.py 4194304 base,color 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:v['ms'] for k,v in d['kernels'].items() if 'ws' in k or 'inference' in k})"
done
echo "######## LDS counters"
cd /tmp && export TMPDIR=/tmp
for v in default swz0; do
  so=""; [ $v = swz0 ] && so=$R/gpurun_abl/lib_swz0.so
  rm -rf /tmp/lds_$v
  LAB4D_SO_PATH=$so timeout -k 5 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d /tmp/lds_$v -- python $R/tools/bench_chain.py 4194304 base,color > /tmp/lds_$v.log 2>&1 || tail -3 /tmp/lds_$v.log
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
done 2>&1 | tee $R/gpurun_out/r06_lds_swizzle.txt
cd $R
echo "######## whole GPU suite"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r06_gpu_tests_call1.txt
echo "######## short bench"
timeout 900 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06_bench_call1.json 2> gpurun_out/r06_bench_call1.err || tail -5 gpurun_out/r06_bench_call1.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r06_bench_call1.json"))
    print("bench", d["value"], d["ms_per_step"], d.get("loss_last_chunk"), d.get("params_finite"), d["config"]["launch"][:30])
    ks = d["roofline"]["kernels_ms_per_step"]
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1])[:14]: print("  ", k, v)
except Exception as e:
    print("bench FAILED", e)
PY
