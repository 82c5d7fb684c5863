"""lab4d_amd -- MI355X-native differentiable volume renderer behind Lab4D's operator API.

The package is a thin Python host layer (autograd plumbing, memory, streams) over
liblab4d_hip.so, a C-ABI library of hand-written gfx950 kernels (include/lab4d_hip.h).
"""
__version__ = "0.1.0"
