// Instantiation of the fused MLP chain kernels for NetSkin18A (see mlp_kernels.hpp).
#include "mlp_kernels.hpp"
LAB4D_MLP_INSTANTIATE(NetSkin18A)
// ... and the fused backward (recompute + dgrad + in-register weight gradients) of this narrow net (mlp_fused_bwd.hpp)
#include "mlp_fused_bwd.hpp"
LAB4D_MLP_INSTANTIATE_FUSED_BWD(NetSkin18A)
