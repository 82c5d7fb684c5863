// Instantiation of the fused MLP chain kernels for NetBgBase (see mlp_kernels.hpp), incl. the tangent-mode forward used by
// the eikonal term of the background field.
#include "mlp_kernels.hpp"
LAB4D_MLP_INSTANTIATE(NetBgBase)

namespace lab4d {
template <>
int launch_mlp_fwd_tangent<NetBgBase>(int precision, const FwdK& k0, int S, hipStream_t st) {
  FwdK k = k0;
  if (precision == LAB4D_PREC_BF16) {
    k.ntiles = k.S_pad / PBF16::TILE;
    LAB4D_MLP_LAUNCH((k_mlp_fwd<NetBgBase, PBF16, true, true>), k, st);
  } else if (precision == LAB4D_PREC_F32) {
    k.ntiles = k.S_pad / PF32::TILE;
    LAB4D_MLP_LAUNCH((k_mlp_fwd<NetBgBase, PF32, true, true>), k, st);
  } else {
    set_error("mlp_forward_tangent: bad precision %d", precision);
    return LAB4D_EINVAL;
  }
  return check_launch("mlp_forward_tangent");
}
}  // namespace lab4d
