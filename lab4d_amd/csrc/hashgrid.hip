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

// Level-major forward (round 6): blockIdx.y = level.  Workgroups are dispatched x-fastest, so at any moment the whole chip works on one or two levels
// and a level's table slab (4 MiB at T = 2^19, F = 2) is what the L2s hold, instead of every wave walking all 16 slabs (64 MiB: every gather a
// trip to the Infinity Cache).  A thread writes its level's F floats of the sample's (L*F)-float row (8 of 128 bytes: partial-sector writes, 10 GB
// per step at the bench's size -- small against the gathers).
template <bool INSIDE_ONLY>
__global__ void __launch_bounds__(256) k_hashgrid_fwd_lm(const float* __restrict__ x, const float* __restrict__ table, const int* __restrict__ res, int S,
                                                         int L, int log2_T, int F, float* __restrict__ out) {
  const size_t slab = ((size_t)1 << log2_T) * F;
  const int l = blockIdx.y;
  const float* tab = table + l * slab;
  const int r = res[l];
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (long)gridDim.x * blockDim.x) {
    const float p[3] = {x[3 * s], x[3 * s + 1], x[3 * s + 2]};
    float f[MAXF];
    if (INSIDE_ONLY && !(p[0] >= 0.f && p[0] <= 1.f && p[1] >= 0.f && p[1] <= 1.f && p[2] >= 0.f && p[2] <= 1.f)) {
      for (int k = 0; k < F; ++k) f[k] = 0.f;
    } else {
      encode_level(p, tab, r, log2_T, F, f);
    }
    for (int k = 0; k < F; ++k) out[(size_t)s * L * F + l * F + k] = f[k];
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

// ---- round 6: the table gradient of the hashed levels through packed 2 x fp16 atomics (F = 2) ----------------------------------------------------
// largest |g_out| of the launch -> *absmax_bits (bits of a non-negative float, ordered like the floats): the scale of the fp16 accumulation
__global__ void __launch_bounds__(256) k_hash_absmax(const float* __restrict__ g, long n, uint32_t* __restrict__ absmax_bits) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float a = fabsf(g[i]);
    m = a > m ? a : m;  // (NaN is skipped: a NaN gradient poisons the fp32 levels and is caught by the optimizer's finite check)
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(absmax_bits, __float_as_uint(m));
}

__global__ void __launch_bounds__(256) k_hashgrid_bwd_h2(const float* __restrict__ x, const float* __restrict__ table, const int* __restrict__ res,
                                                         const float* __restrict__ g_out, int S, int L, int log2_T, int l16, float* __restrict__ g_table,
                                                         uint32_t* __restrict__ g16, const uint32_t* __restrict__ absmax_bits, float* __restrict__ g_x) {
  constexpr int F = 2;
  const size_t slab = ((size_t)1 << log2_T) * F, slab16 = (size_t)1 << log2_T;
  const int lane = threadIdx.x & 63;
  const float scale = h2_scale_of(*absmax_bits);
  for (long s0 = (long)blockIdx.x * blockDim.x; s0 < S; s0 += (long)gridDim.x * blockDim.x) {
    const long s = s0 + threadIdx.x;
    const bool live = s < S;
    const long sc = live ? s : S - 1;
    const float p[3] = {x[3 * sc], x[3 * sc + 1], x[3 * sc + 2]};
    float gx[3] = {0.f, 0.f, 0.f};
    for (int l = 0; l < L; ++l) {
      float g[2];
      g[0] = live ? g_out[(size_t)s * L * F + l * F] : 0.f;
      g[1] = live ? g_out[(size_t)s * L * F + l * F + 1] : 0.f;
      if (!__any(g[0] != 0.f || g[1] != 0.f)) continue;
      if (l < l16) encode_level_bwd<true>(p, table + l * slab, res[l], log2_T, F, g, g_table + l * slab, g_x ? gx : nullptr, lane);
      else encode_level_bwd<true>(p, table + l * slab, res[l], log2_T, F, g, nullptr, g_x ? gx : nullptr, lane, g16 + l * slab16, scale);
    }
    if (g_x && live) { g_x[3 * s] = gx[0]; g_x[3 * s + 1] = gx[1]; g_x[3 * s + 2] = gx[2]; }
  }
}

// g_table[l][v][f] += fp16 word of (l, v) / scale for the levels l >= l16; the words are cleared for the next launch
__global__ void __launch_bounds__(256) k_hash_flush_h2(uint32_t* __restrict__ g16, const uint32_t* __restrict__ absmax_bits, long w0, long n_words,
                                                       float* __restrict__ g_table) {
  const float inv = 1.f / h2_scale_of(*absmax_bits);
  for (long i = w0 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (long)gridDim.x * blockDim.x) {
    const uint32_t w = g16[i];
    if (w == 0u) continue;
    lab4d_h2 h;
    __builtin_memcpy(&h, &w, 4);
    g_table[2 * i] += (float)h[0] * inv;
    g_table[2 * i + 1] += (float)h[1] * inv;
    g16[i] = 0u;
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
  static const int lm = getenv("LAB4D_HASH_LEVEL_MAJOR") ? atoi(getenv("LAB4D_HASH_LEVEL_MAJOR")) : 1;  // 0: the sample-major kernel (A/B measurements)
  if (lm && S >= 65536) {
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_hashgrid_fwd_lm<true>, dim3((int)g, L), dim3(256), 0, (hipStream_t)stream, x, table, res, S, L, log2_T, F, out);
    return check_launch("hashgrid_forward_inside");
  }
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

extern "C" int lab4d_hashgrid_absmax(const float* g_out, long n, uint32_t* absmax_bits, void* stream) {
  LAB4D_REQUIRE(g_out && absmax_bits && n >= 0, "hashgrid_absmax: bad arguments");
  if (n == 0) return LAB4D_OK;
  long g = (n + 255L) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_hash_absmax, dim3((int)g), dim3(256), 0, (hipStream_t)stream, g_out, n, absmax_bits);
  return check_launch("hashgrid_absmax");
}

extern "C" int lab4d_hashgrid_backward_f16(const float* x, const float* table, const int32_t* res, const float* g_out, int S, int L, int log2_T,
                                           int first_f16_level, float* g_table, uint32_t* g16, const uint32_t* absmax_bits, float* g_x, void* stream) {
  const int F = 2;
  HASH_CHECKS("hashgrid_backward_f16");
  LAB4D_REQUIRE(g_out && g_table && g16 && absmax_bits, "hashgrid_backward_f16: null pointer");
  LAB4D_REQUIRE(first_f16_level >= 0 && first_f16_level <= L, "hashgrid_backward_f16: first_f16_level %d outside 0..%d", first_f16_level, L);
  long g = (S + 255L) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(k_hashgrid_bwd_h2, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, table, res, g_out, S, L, log2_T, first_f16_level, g_table, g16,
                     absmax_bits, g_x);
  return check_launch("hashgrid_backward_f16");
}

extern "C" int lab4d_hashgrid_flush_f16(uint32_t* g16, const uint32_t* absmax_bits, int L, int log2_T, int first_f16_level, float* g_table, void* stream) {
  LAB4D_REQUIRE(g16 && absmax_bits && g_table, "hashgrid_flush_f16: null pointer");
  LAB4D_REQUIRE(L > 0 && L <= 32 && log2_T >= 4 && log2_T <= 24 && first_f16_level >= 0 && first_f16_level <= L, "hashgrid_flush_f16: bad sizes");
  if (first_f16_level == L) return LAB4D_OK;
  const long T = 1L << log2_T, w0 = (long)first_f16_level * T, n = (long)L * T;
  long g = (n - w0 + 255L) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_hash_flush_h2, dim3((int)g), dim3(256), 0, (hipStream_t)stream, g16, absmax_bits, w0, n, g_table);
  return check_launch("hashgrid_flush_f16");
}
