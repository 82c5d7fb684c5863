#!/bin/bash
# Compile the measurement probes for gfx950 (binaries are git-ignored but travel to the GPU box with the snapshot):
#   tools/probes/build.sh && gpurun -- 'tools/probes/mfma_core.bin; tools/probes/store_ack.bin'     (each runs in a few seconds)
cd "$(dirname "$0")"
for p in mfma_core mfma_core2 store_ack store_roof tr_probe ws_core ring_probe; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $p.bin $p.hip 2>&1 | grep -E "error" -A3
done
ls -la *.bin
