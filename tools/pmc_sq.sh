#!/bin/bash
# SQ counters for the chain kernels (separate passes per counter group). usage: pmc_sq.sh "<cmd>" "<kernel regex>"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD=${1:-"python $R/tools/bench_mlp_fwd.py 2097152"}
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/sq_$i -- $CMD > /tmp/sq_$i.log 2>&1 || tail -5 /tmp/sq_$i.log
done
python - <<'PY'
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob('/tmp/sq_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if "k_mlp_wgrad" not in k: continue
        k = re.sub(r'lab4d::', '', k.split('(')[0])[:70]
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print('   %-28s n=%d  max=%.4g  mean=%.4g' % (c, len(v), max(v), sum(v)/len(v)))
PY
