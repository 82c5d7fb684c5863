#!/bin/bash
# SQ counters (MFMA-busy, VALU, wait cycles) for the MLP kernels, two separate --pmc passes (never combined with other
# trace domains).  usage: pmc_sq.sh ["<cmd>"]   -> gpurun_out/r01_sq_counters.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD=${1:-"python $R/tools/bench_mlp.py 2097152 1"}
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 5 100 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/sq_$i -- $CMD > /tmp/sq_$i.log 2>&1 || tail -5 /tmp/sq_$i.log
done
mkdir -p $R/gpurun_out
python - > $R/gpurun_out/r01_sq_counters.txt <<'PY'
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob('/tmp/sq_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'k_mlp' not in k: continue
        k = re.sub(r'lab4d::', '', k.split('(')[0])[:70]
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
print("# rocprofv3 --pmc (two passes) over tools/bench_mlp.py 2097152 1 (basefield + colourfield chains + wgrad, 2.1 M samples)")
print("# per launch means.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (cycles); GRBM_GUI_ACTIVE over the 8 XCDs;")
print("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)")
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    line = "%-62s" % k + " ".join("%s=%.4g" % (n, c[n]) for n in sorted(c))
    if 'GRBM_GUI_ACTIVE' in c and 'SQ_VALU_MFMA_BUSY_CYCLES' in c and c['GRBM_GUI_ACTIVE'] > 0:
        line += "  mfma_util=%.3f" % (c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0))
    print(line)
PY
cat $R/gpurun_out/r01_sq_counters.txt | cut -c1-260
