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

// Tangent input of the eikonal term's backward (NeRF.compute_eikonal, nerf.py:416-453, through mlp.EikonalSdf): with g = d sdf / dx, |g| = gn and the
// incoming gradient ge of (gn - 1)^2:  dL/dg = ge * 2 (gn - 1) / gn * g  (0 where gn == 0: the zero subgradient torch's norm backward takes), and
// u = J_e(x) dL/dg in embedding-slot order: slot 2(3f + a) = w_f 2^f cos(2^f x_a) dLdg_a, slot 2(3f + a) + 1 = -w_f 2^f sin(2^f x_a) dLdg_a, then dLdg (3), then zeros
// up to KE.  One thread per sample; replaces a dozen element-wise torch launches over (S, 6L) tensors.
__global__ void __launch_bounds__(256) k_eik_tangent_input(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ ge,
                                                           const float* __restrict__ freq_w, long S, int L, int KE, float* __restrict__ u) {
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (long)gridDim.x * blockDim.x) {
    const float g0 = g[s * 3], g1 = g[s * 3 + 1], g2 = g[s * 3 + 2];
    const float gn = sqrtf(g0 * g0 + g1 * g1 + g2 * g2);
    const float k = gn > 0.f ? ge[s] * (2.f * (gn - 1.f) / fmaxf(gn, 1e-38f)) : 0.f;
    const float d[3] = {k * g0, k * g1, k * g2};
    const float xv[3] = {x[s * 3], x[s * 3 + 1], x[s * 3 + 2]};
    float* row = u + s * KE;
    for (int f = 0; f < L; ++f) {
      const float fr = ldexpf(1.f, f);
      const float wf = freq_w ? fr * freq_w[f] : fr;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        float sn, cs;
        sincosf(xv[a] * fr, &sn, &cs);
        row[2 * (3 * f + a)] = wf * cs * d[a];
        row[2 * (3 * f + a) + 1] = -wf * sn * d[a];
      }
    }
    row[6 * L] = d[0]; row[6 * L + 1] = d[1]; row[6 * L + 2] = d[2];
    for (int c = 6 * L + 3; c < KE; ++c) row[c] = 0.f;
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

extern "C" int lab4d_eikonal_tangent_input(const float* x, const float* g, const float* ge, const float* freq_w, int S, int L, int KE, float* u,
                                           void* stream) {
  LAB4D_REQUIRE(x && g && ge && u, "eikonal_tangent_input: null pointer");
  LAB4D_REQUIRE(L >= 0 && KE >= 6 * L + 3, "eikonal_tangent_input: KE = %d is too small for %d bands", KE, L);
  if (S == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_eik_tangent_input, dim3(pw_grid(S)), dim3(256), 0, (hipStream_t)stream, x, g, ge, freq_w, (long)S, L, KE, u);
  return check_launch("eikonal_tangent_input");
}
