// Instantiation of the fused MLP chain kernels for the hash-grid field nets (see mlp_kernels.hpp, mlp_nets.hpp).
#include "mlp_kernels.hpp"
LAB4D_MLP_INSTANTIATE(NetHashGeo)
LAB4D_MLP_INSTANTIATE(NetHashColor)
