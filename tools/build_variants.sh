#!/bin/bash
# Build kernel-experiment variants of liblab4d_hip.so: one per -DLAB4D_ABL_<NAME> (BASE = no define) into gpurun_abl/.
# usage: tools/build_variants.sh BASE NOA ...   then   LAB4D_SO_PATH=gpurun_abl/lib_NOA.so python tools/bench_mlp_fwd.py
cd "$(dirname "$0")/.."
mkdir -p gpurun_abl
for v in "$@"; do
  if [ "$v" = BASE ]; then X=""; else X="-DLAB4D_ABL_$v"; fi
  LAB4D_HIPCC_EXTRA="$X" LAB4D_SO_PATH=$PWD/gpurun_abl/lib_$v.so LAB4D_BUILD_DIR=/tmp/build_$v \
    python -c "from lab4d_amd import _lib; _lib.build(verbose=False)" || echo "FAIL $v" &
done
wait
ls -la gpurun_abl/
