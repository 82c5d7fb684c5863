// Instantiation of the fused MLP chain kernels for NetDense6 (fg_motion "dense"; see mlp_kernels.hpp).
#include "mlp_kernels.hpp"
LAB4D_MLP_INSTANTIATE(NetDense6)
