#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
for v in lab4d_amd/liblab4d_hip.so gpurun_abl/lib_*.so; do echo "## $v"; LAB4D_SO_PATH=$R/$v timeout 100 python tools/ws_compare.py --nets fg_base --quick --time 4194304 2>&1 | grep '"net"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:v for k,v in d.items() if k.endswith('_ms')})"; done
