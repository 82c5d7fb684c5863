// Instantiation of the fused MLP chain kernels for NetFgBase (see mlp_kernels.hpp), incl. the tangent-mode forward used by
// the eikonal term.
#include "mlp_kernels.hpp"
LAB4D_MLP_INSTANTIATE(NetFgBase)

namespace lab4d {
template <>
int launch_mlp_fwd_tangent<NetFgBase>(int precision, const FwdK& k0, int S, hipStream_t st) {
  FwdK k = k0;
  if (precision == LAB4D_PREC_BF16) {
    if (ws_enabled()) {  // the weights-stationary family (mlp_kernels_ws.hpp), bit-equal
      hipLaunchKernelGGL((k_mlp_fwd_ws<NetFgBase, true, true, true>), dim3(mlp_grid_ws(k.S_pad / WS_TILE)), dim3(512), 0, st, k);
      return check_launch("mlp_forward_tangent");
    }
    k.ntiles = k.S_pad / PBF16::TILE;
    LAB4D_MLP_LAUNCH((k_mlp_fwd<NetFgBase, PBF16, true, true>), k, st);
  } else if (precision == LAB4D_PREC_F32) {
    k.ntiles = k.S_pad / PF32::TILE;
    LAB4D_MLP_LAUNCH((k_mlp_fwd<NetFgBase, PF32, true, true>), k, st);
  } else {
    set_error("mlp_forward_tangent: bad precision %d", precision);
    return LAB4D_EINVAL;
  }
  return check_launch("mlp_forward_tangent");
}
}  // namespace lab4d
