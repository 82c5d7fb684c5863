// Fused per-sample epilogues of the field that the reference writes as chains of element-wise torch ops.
// Contract: include/lab4d_hip.h (section 3c).
#include "common.hpp"

namespace lab4d {

// FeatureNeRF.compute_feat (lab4d/nnutils/feature.py:149-150): y = f / ||f||_2 over the last axis (C <= 32 channels, no eps).
// One thread per sample; the reference's norm + div (+ 6 kernels in the backward) each stream the (S,C) tensor again.
template <int C>
__global__ void __launch_bounds__(256) k_l2norm_fwd(const float* __restrict__ x, long S, float* __restrict__ y) {
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (long)gridDim.x * blockDim.x) {
    float v[C];
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { v[c] = x[s * C + c]; q += v[c] * v[c]; }
    const float inv = 1.f / sqrtf(q);
#pragma unroll
    for (int c = 0; c < C; ++c) y[s * C + c] = v[c] * inv;
  }
}
// g_x = (g - y (y . g)) / ||x||
template <int C>
__global__ void __launch_bounds__(256) k_l2norm_bwd(const float* __restrict__ x, const float* __restrict__ g, long S, float* __restrict__ gx) {
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (long)gridDim.x * blockDim.x) {
    float v[C], gv[C];
    float q = 0.f, d = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { v[c] = x[s * C + c]; gv[c] = g[s * C + c]; q += v[c] * v[c]; d += v[c] * gv[c]; }
    const float inv = 1.f / sqrtf(q);
    const float k = d * inv * inv;  // (y . g) / ||x|| = (x . g) / ||x||^2 ... times y = x / ||x||
#pragma unroll
    for (int c = 0; c < C; ++c) gx[s * C + c] = (gv[c] - v[c] * k) * inv;
  }
}

inline int pw_grid(long S) {
  long g = (S + 255) / 256;
  return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace lab4d
using namespace lab4d;

extern "C" int lab4d_l2_normalize_forward(const float* x, int S, int C, float* y, void* stream) {
  LAB4D_REQUIRE(x && y, "l2_normalize_forward: null pointer");
  LAB4D_REQUIRE(C == 16 || C == 3, "l2_normalize_forward: C must be 16 (feature field) or 3 (got %d)", C);
  if (S == 0) return LAB4D_OK;
  if (C == 16) hipLaunchKernelGGL((k_l2norm_fwd<16>), dim3(pw_grid(S)), dim3(256), 0, (hipStream_t)stream, x, (long)S, y);
  else hipLaunchKernelGGL((k_l2norm_fwd<3>), dim3(pw_grid(S)), dim3(256), 0, (hipStream_t)stream, x, (long)S, y);
  return check_launch("l2_normalize_forward");
}

extern "C" int lab4d_l2_normalize_backward(const float* x, const float* g, int S, int C, float* g_x, void* stream) {
  LAB4D_REQUIRE(x && g && g_x, "l2_normalize_backward: null pointer");
  LAB4D_REQUIRE(C == 16 || C == 3, "l2_normalize_backward: C must be 16 or 3 (got %d)", C);
  if (S == 0) return LAB4D_OK;
  if (C == 16) hipLaunchKernelGGL((k_l2norm_bwd<16>), dim3(pw_grid(S)), dim3(256), 0, (hipStream_t)stream, x, g, (long)S, g_x);
  else hipLaunchKernelGGL((k_l2norm_bwd<3>), dim3(pw_grid(S)), dim3(256), 0, (hipStream_t)stream, x, g, (long)S, g_x);
  return check_launch("l2_normalize_backward");
}
