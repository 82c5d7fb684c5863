#!/bin/bash
# Build kernel-experiment variants of liblab4d_hip.so into gpurun_abl/: one per argument; BASE = no define, NAME = -DLAB4D_ABL_NAME,
# A+B = both defines; a name starting with '=' is taken literally (=A_NT -> -DLAB4D_A_NT).
#   usage: tools/build_variants.sh BASE NOSTORE NOSTORE+NOMASK =A_NT =A_SC =TRSPREAD =SCHED_IL ...
#   then on the GPU box:   for v in gpurun_abl/lib_*.so; do LAB4D_SO_PATH=$v python tools/bench_chain.py; done
# Timing-only ablations (results wrong): NOSTORE NOMASK NOAFETCH NOBAR.  Correct builds: MASK1 (one-step sign-word prefetch), NOPROG,
# OCC1, ACG14/ACG7, WGRAD4, PLAINSTORE, =A_NT / =A_SC (weight loads nt / sc0 sc1), =TRSPREAD (backward dZ tiles leave the slab in pieces),
# =SCHED_IL (MFMA / VALU interleave pattern), =TRSTORE (transposing-read tile stores), =ACACHE_G=16 (all of the LDS for shared weights),
# =ST_AGPR / =ST_BUF (tile stores with AGPR data / as buffer_store with an SGPR base: the store-form experiments of DESIGN.md section 8).
cd "$(dirname "$0")/.."
mkdir -p gpurun_abl
for v in "$@"; do
  X=""
  if [ "$v" != BASE ]; then for d in ${v//+/ }; do if [ "${d:0:1}" = "=" ]; then X="$X -DLAB4D_${d:1}"; else X="$X -DLAB4D_ABL_$d"; fi; done; fi
  LAB4D_HIPCC_EXTRA="$X" LAB4D_SO_PATH=$PWD/gpurun_abl/lib_${v//=/}.so LAB4D_BUILD_DIR=/tmp/build_${v//=/} \
    python -c "from lab4d_amd import _lib; _lib.build(verbose=False)" || echo "FAIL $v"
done
ls -la gpurun_abl/
