// Instantiation of the fused MLP chain kernels for NetSkin18A (see mlp_kernels.hpp).
#include "mlp_kernels.hpp"
LAB4D_MLP_INSTANTIATE(NetSkin18A)
