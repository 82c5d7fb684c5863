"""lab4d_amd -- MI355X-native differentiable volume renderer behind Lab4D's operator API.

The package is a thin Python host layer (autograd plumbing, memory, streams) over
liblab4d_hip.so, a C-ABI library of hand-written gfx950 kernels (include/lab4d_hip.h).
"""
__version__ = "0.1.0"

import os as _os

# hipGraph replay on ROCm 7.x: with the runtime's AQL packet capture (the default) a captured hipMemsetAsync node is not ordered against the
# kernel nodes around it -- the first replay runs on fresh (zero) pool memory and looks right, later replays accumulate onto stale sums.
# The kernels of this library clear their accumulators with a fill kernel (csrc/common.hpp zero_async), but PyTorch's own multi-block
# reductions memset their semaphores, so a captured training chunk that contains one (the gradient of a broadcast scalar, t_sum.sum())
# went wrong from the second replay on (round 3, bench.py).  The flag is read when the HIP runtime initialises: it must be in the
# environment before the first device call, so it is set here on import (and at the top of bench.py / tests/conftest.py); measured cost
# on the bench: none (836.8 vs 837.2 ms per step).
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
