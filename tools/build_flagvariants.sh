#!/bin/bash
# name:flags ...
cd /root/repo
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  LAB4D_HIPCC_EXTRA="$f" LAB4D_SO_PATH=$PWD/gpurun_abl/lib_$n.so LAB4D_BUILD_DIR=/tmp/build_$n python -c "from lab4d_amd import _lib; _lib.build(verbose=False)" || echo "FAIL $n"
done
ls -la gpurun_abl/
