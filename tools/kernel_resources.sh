#!/bin/bash
# Static resources of every chain-kernel instantiation (VGPR / AGPR / scratch / LDS / occupancy), from -Rpass-analysis.
#   tools/kernel_resources.sh [files...]   (default: every mlp_inst_*.hip)     extra flags via LAB4D_HIPCC_EXTRA
cd "$(dirname "$0")/.."
FILES=${@:-$(ls lab4d_amd/csrc/mlp_inst_*.hip)}
for f in $FILES; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ilab4d_amd/csrc -Wno-unused-value -Wno-pass-failed $LAB4D_HIPCC_EXTRA \
    -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 | python3 -c "
import sys,re,subprocess
name=None;d={}
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m:
        name=subprocess.run(['/usr/bin/c++filt',m.group(1)],capture_output=True,text=True).stdout.strip(); d={}
    for key in [' VGPRs:',' AGPRs:','ScratchSize','Occupancy','LDS Size','VGPRs Spill']:
        if key in line and name:
            d[key.strip()]=re.findall(r': (\d+) \[-Rpass',line)[0]
    if 'LDS Size' in line and name:
        print('%-70s V %s A %s scratch %s spill %s LDS %s occ %s'%(name.replace('lab4d::','')[:70],d.get('VGPRs:'),d.get('AGPRs:'),d.get('ScratchSize'),d.get('VGPRs Spill'),d.get('LDS Size'),d.get('Occupancy')))
"
done
