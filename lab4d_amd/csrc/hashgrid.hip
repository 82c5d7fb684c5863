// Multiresolution hash encoding on gfx950 (BASELINE config 5; no counterpart in the reference: SURVEY F3, parity unpinned).
// Contract: include/lab4d_hashgrid.h; arithmetic: hashgrid_math.hpp.  HBM / Infinity-Cache-bound gather: per sample L levels x 8
// vertices x F floats of random reads (1 KiB at L=16, F=2) and L*F floats written.  One thread per sample walks the levels, so
// the (S, L*F) output row is written by one lane (128 B contiguous at L*F = 32) and d/dx needs no cross-thread reduction; the
// table gradient goes through fp32 atomics (vertices of fine levels are hit by few samples, coarse levels stay L2-resident).
// A level-major launch (blockIdx.y = level) that keeps one level's table slice hot per XCD is the variant to measure next.
#include "common.hpp"
#include "hashgrid_math.hpp"

namespace lab4d {
using namespace lab4d_hash;

template <bool INSIDE_ONLY>
__global__ void __launch_bounds__(256) k_hashgrid_fwd(const float* __restrict__ x, const float* __restrict__ table, const int* __restrict__ res, int S,
                                                      int L, int log2_T, int F, float* __restrict__ out) {
  const size_t slab = ((size_t)1 << log2_T) * F;
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (long)gridDim.x * blockDim.x) {
    const float p[3] = {x[3 * s], x[3 * s + 1], x[3 * s + 2]};
    if (INSIDE_ONLY && !(p[0] >= 0.f && p[0] <= 1.f && p[1] >= 0.f && p[1] <= 1.f && p[2] >= 0.f && p[2] <= 1.f)) {  // (NaN: outside)
      for (int k = 0; k < L * F; ++k) out[(size_t)s * L * F + k] = 0.f;
      continue;
    }
    for (int l = 0; l < L; ++l) {
      float f[MAXF];
      encode_level(p, table + l * slab, res[l], log2_T, F, f);
      for (int k = 0; k < F; ++k) out[(size_t)s * L * F + l * F + k] = f[k];
    }
  }
}

__global__ void __launch_bounds__(256) k_hashgrid_bwd(const float* __restrict__ x, const float* __restrict__ table, const int* __restrict__ res,
                                                      const float* __restrict__ g_out, int S, int L, int log2_T, int F, float* __restrict__ g_table,
                                                      float* __restrict__ g_x) {
  const size_t slab = ((size_t)1 << log2_T) * F;
  const int lane = threadIdx.x & 63;
  // wave-uniform trip count: the table updates are combined across the lanes of a wave (wave_run_add), so every lane takes part;
  // lanes past the end carry a zero gradient
  for (long s0 = (long)blockIdx.x * blockDim.x; s0 < S; s0 += (long)gridDim.x * blockDim.x) {
    const long s = s0 + threadIdx.x;
    const bool live = s < S;
    const long sc = live ? s : S - 1;
    const float p[3] = {x[3 * sc], x[3 * sc + 1], x[3 * sc + 2]};
    float gx[3] = {0.f, 0.f, 0.f};
    for (int l = 0; l < L; ++l) {
      float g[MAXF];
      bool nz = false;
      for (int k = 0; k < F; ++k) {
        g[k] = live ? g_out[(size_t)s * L * F + l * F + k] : 0.f;
        nz = nz || g[k] != 0.f;
      }
      if (!__any(nz)) continue;  // wave-uniform: nothing to add for these 64 samples at this level (masked samples of a box-only field)
      encode_level_bwd<true>(p, table + l * slab, res[l], log2_T, F, g, g_table ? g_table + l * slab : nullptr, g_x ? gx : nullptr, lane);
    }
    if (g_x && live) { g_x[3 * s] = gx[0]; g_x[3 * s + 1] = gx[1]; g_x[3 * s + 2] = gx[2]; }
  }
}

}  // namespace lab4d

using namespace lab4d;

#define HASH_CHECKS(name)                                                                                                              \
  LAB4D_REQUIRE(x && table && res, name ": null pointer");                                                                              \
  LAB4D_REQUIRE(S >= 0 && L > 0 && L <= 32 && log2_T >= 4 && log2_T <= 24 && F >= 1 && F <= lab4d_hash::MAXF, name ": bad sizes S=%d L=%d log2_T=%d F=%d", \
                S, L, log2_T, F);                                                                                                       \
  if (S == 0) return LAB4D_OK;

extern "C" int lab4d_hashgrid_forward(const float* x, const float* table, const int32_t* res, int S, int L, int log2_T, int F, float* out,
                                      void* stream) {
  HASH_CHECKS("hashgrid_forward");
  LAB4D_REQUIRE(out, "hashgrid_forward: null output");
  long g = (S + 255L) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(k_hashgrid_fwd<false>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, table, res, S, L, log2_T, F, out);
  return check_launch("hashgrid_forward");
}

extern "C" int lab4d_hashgrid_forward_inside(const float* x, const float* table, const int32_t* res, int S, int L, int log2_T, int F, float* out,
                                             void* stream) {
  HASH_CHECKS("hashgrid_forward_inside");
  LAB4D_REQUIRE(out, "hashgrid_forward_inside: null output");
  long g = (S + 255L) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(k_hashgrid_fwd<true>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, table, res, S, L, log2_T, F, out);
  return check_launch("hashgrid_forward_inside");
}

extern "C" int lab4d_hashgrid_backward(const float* x, const float* table, const int32_t* res, const float* g_out, int S, int L, int log2_T, int F,
                                       float* g_table, float* g_x, void* stream) {
  HASH_CHECKS("hashgrid_backward");
  LAB4D_REQUIRE(g_out && (g_table || g_x), "hashgrid_backward: null pointer");
  long g = (S + 255L) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(k_hashgrid_bwd, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, table, res, g_out, S, L, log2_T, F, g_table, g_x);
  return check_launch("hashgrid_backward");
}
