#!/bin/bash
# Build kernel-experiment variants of liblab4d_hip.so into gpurun_abl/: one per argument; BASE = no define, NAME = -DLAB4D_ABL_NAME,
# A+B = both defines.   usage: tools/build_variants.sh BASE NOSTORE NOSTORE+NOMASK ...
#   then on the GPU box:   for v in gpurun_abl/lib_*.so; do LAB4D_SO_PATH=$v python tools/bench_chain.py; done
cd "$(dirname "$0")/.."
mkdir -p gpurun_abl
for v in "$@"; do
  X=""
  if [ "$v" != BASE ]; then for d in ${v//+/ }; do X="$X -DLAB4D_ABL_$d"; done; fi
  LAB4D_HIPCC_EXTRA="$X" LAB4D_SO_PATH=$PWD/gpurun_abl/lib_$v.so LAB4D_BUILD_DIR=/tmp/build_$v \
    python -c "from lab4d_amd import _lib; _lib.build(verbose=False)" || echo "FAIL $v"
done
ls -la gpurun_abl/
