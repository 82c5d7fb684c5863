#!/bin/bash
# Memory-path stall counters of the chain kernels (separate --pmc passes, kernel-trace only) -> gpurun_out/r02_stall_counters.txt
# Names from `rocprofv3 -L` on the MI355X box (profiles/r02_counter_list.txt).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD=${1:-"python $R/tools/bench_chain.py 2097152 base"}
i=0
for grp in "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
           "TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
           "TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum" \
           "TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 60 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/st_$i -- $CMD > /tmp/st_$i.log 2>&1 || { echo "group $i failed: $grp"; tail -3 /tmp/st_$i.log; }
done
python - > $R/gpurun_out/r02_stall_counters.txt <<'PY'
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob('/tmp/st_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'k_mlp_fwd' not in k and 'k_mlp_bwd' not in k and 'k_mlp_wgrad' not in k: continue
        k = re.sub(r'lab4d::', '', k.split('(')[0])[:60]
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    print(k)
    for n in sorted(c):
        print("   %-46s %.4g" % (n, c[n]))
PY
cat $R/gpurun_out/r02_stall_counters.txt
