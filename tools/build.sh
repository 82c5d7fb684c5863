#!/bin/bash
# Build liblab4d_hip.so in-tree (from any cwd) and report.
cd "$(dirname "$0")/.." && python -c "from lab4d_amd import _lib; _lib.build(verbose=True)" 2>&1 | grep -v "^\[lab4d_amd\] compiled" | tail -${1:-15}; ls -la lab4d_amd/liblab4d_hip.so
