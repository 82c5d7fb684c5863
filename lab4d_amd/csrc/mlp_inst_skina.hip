// Instantiation of the fused MLP chain kernels for NetSkinA (see mlp_kernels.hpp).
#include "mlp_kernels.hpp"
LAB4D_MLP_INSTANTIATE(NetSkinA)
