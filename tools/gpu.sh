#!/bin/bash
# Rebuild the in-tree library (a stale .so is what travels to the GPU box otherwise), then run a command on an MI355X.
#   tools/gpu.sh <timeout-seconds> '<command>'
cd "$(dirname "$0")/.."
python -c 'import __graft_entry__ as g; g.build()' 2>&1 | tail -1 || exit 1
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
