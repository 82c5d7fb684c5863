#!/bin/bash
# SQ wait / issue breakdown of the chain kernels (separate --pmc passes, kernel-trace only).  -> gpurun_out/${SQ_OUT:-r04_sq_counters.txt}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD=${1:-"python $R/tools/bench_chain.py 2097152 base"}
[ -n "$SQ_SKIP_LIST" ] || timeout -k 5 60 rocprofv3 -L > $R/gpurun_out/r05_counter_list.txt 2>&1  # every counter this box offers
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_WAVE32_LDS" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/sq2_$i -- $CMD > /tmp/sq2_$i.log 2>&1 || tail -3 /tmp/sq2_$i.log
done
python - > $R/gpurun_out/${SQ_OUT:-r04_sq_counters.txt} <<'PY'
import csv, glob, os, re
FILTER = os.environ.get('SQ_FILTER', 'k_mlp_fwd|k_mlp_bwd')  # kernels of interest (regex on the symbol)
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob('/tmp/sq2_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if not re.search(FILTER, k): continue
        k = re.sub(r'lab4d::', '', k.split('(')[0])[:60]
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    print(k)
    wc = c.get('SQ_WAVE_CYCLES', 0)
    for n in sorted(c):
        print("   %-28s %.4g %s" % (n, c[n], ("(%.1f%% of wave cycles)" % (100 * c[n] / wc)) if wc and n.startswith(('SQ_WAIT', 'SQ_ACTIVE')) else ""))
PY
cat $R/gpurun_out/${SQ_OUT:-r04_sq_counters.txt}
