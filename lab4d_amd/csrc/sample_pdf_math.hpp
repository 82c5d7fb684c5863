// Per-ray arithmetic of sample_pdf (lab4d/utils/render_utils.py:187-233) in the reference's own order of floating-point operations.
// Plain C++ behind LAB4D_HD (see fk_math.hpp): csrc/raymarch.hip runs it one thread per ray, tests/host_harness/sample_pdf_host.cpp
// compiles it with g++ (-ffp-contract=off) so that the CPU test-suite can hold it bit for bit to torch's CPU kernels.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define LAB4D_HD __host__ __device__ inline
#else
#define LAB4D_HD inline
#endif

namespace lab4d_pdf {

// rounded fp32 sum / product / quotient that must not be contracted into an fma (device: made opaque to the optimiser, see common.hpp)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float add_rn(float a, float b) { float p = a + b; asm volatile("" : "+v"(p)); return p; }
__device__ __forceinline__ float mul_rn(float a, float b) { float p = a * b; asm volatile("" : "+v"(p)); return p; }
__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }
#else
inline float add_rn(float a, float b) { volatile float p = a + b; return p; }
inline float mul_rn(float a, float b) { volatile float p = a * b; return p; }
inline float div_rn(float a, float b) { volatile float p = a / b; return p; }
#endif

// torch.sum(x, -1) of a contiguous fp32 row on the CPU, bit for bit: ATen's cascade_sum / vectorized_inner_sum
// (aten/src/ATen/native/cpu/SumKernel.cpp) keeps 4 independent vector accumulators (ILP) of 8 float lanes (the kernel is built for AVX2 and
// dispatched as such on AVX512 hosts too; checked on both capabilities in tests/test_oracle_properties.py), folds rows of 4 vectors into
// them with a 4-level cascade of 16-row blocks, adds the left-over whole vectors to accumulator 0, folds accumulators 1..3 into 0, then
// adds the scalar tail and the 8 lanes sequentially.  Rows shorter than one vector take the same scheme with scalar "lanes".
// x_i = w[i] + eps (the reference's `weights + eps`, render_utils.py:203).  The normaliser decides whether cdf[-1] lands below, on or
// above 1.0f, i.e. which bin searchsorted(right=True) returns for u = 1 -- so the ORDER of these additions is part of the result.
template <int L>
LAB4D_HD float torch_cpu_row_sum(const float* __restrict__ w, int n, float eps) {
  const int vs = n / L;              // whole "vectors" (L = 8 lanes, or 1 for the scalar scheme)
  const int size_ilp = vs / 4;       // rows of 4 vectors
  float ps[4][L];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int l = 0; l < L; ++l) ps[k][l] = 0.f;
  if (size_ilp < 16) {               // one cascade block at most: plain sequential accumulation (every ray of this renderer: n <= 254)
    for (int i = 0; i < size_ilp; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < L; ++l) ps[k][l] = add_rn(ps[k][l], add_rn(w[(4 * i + k) * L + l], eps));
  } else {                           // multi_row_sum's cascade: level j holds the sum of 16^j-row blocks
    int lp = 0;
    while ((1 << lp) < size_ilp) ++lp;  // CeilLog2(size)
    lp = lp / 4 > 4 ? lp / 4 : 4;
    const int level_step = 1 << lp, level_mask = level_step - 1;
    float acc[3][4][L];                 // levels 1..3 (level 0 is ps)
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 4; ++k)
        for (int l = 0; l < L; ++l) acc[j][k][l] = 0.f;
    int i = 0;
    while (i + level_step <= size_ilp) {
      for (int j = 0; j < level_step; ++j, ++i)
        for (int k = 0; k < 4; ++k)
          for (int l = 0; l < L; ++l) ps[k][l] = add_rn(ps[k][l], add_rn(w[(4 * i + k) * L + l], eps));
      for (int j = 1; j < 4; ++j) {
        for (int k = 0; k < 4; ++k)
          for (int l = 0; l < L; ++l) {
            acc[j - 1][k][l] = add_rn(acc[j - 1][k][l], j == 1 ? ps[k][l] : acc[j - 2][k][l]);
            if (j == 1) ps[k][l] = 0.f; else acc[j - 2][k][l] = 0.f;
          }
        if ((i & (level_mask << (j * lp))) != 0) break;
      }
    }
    for (; i < size_ilp; ++i)
      for (int k = 0; k < 4; ++k)
        for (int l = 0; l < L; ++l) ps[k][l] = add_rn(ps[k][l], add_rn(w[(4 * i + k) * L + l], eps));
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 4; ++k)
        for (int l = 0; l < L; ++l) ps[k][l] = add_rn(ps[k][l], acc[j][k][l]);
  }
  for (int i = size_ilp * 4; i < vs; ++i)
#pragma unroll
    for (int l = 0; l < L; ++l) ps[0][l] = add_rn(ps[0][l], add_rn(w[i * L + l], eps));
#pragma unroll
  for (int k = 1; k < 4; ++k)
#pragma unroll
    for (int l = 0; l < L; ++l) ps[0][l] = add_rn(ps[0][l], ps[k][l]);
  if (L == 1) return ps[0][0];       // scalar scheme: row_sum's result is the sum
  float fin = 0.f;
  for (int k = vs * L; k < n; ++k) fin = add_rn(fin, add_rn(w[k], eps));
#pragma unroll
  for (int l = 0; l < L; ++l) fin = add_rn(fin, ps[0][l]);
  return fin;
}

// One ray of sample_pdf: bins (n_w + 1), weights (n_w) -> samples (n_imp) and the searchsorted(right=True) indices (n_imp).
// u_in: the ray's n_imp queries sorted ascending (det=False; the caller un-sorts), or NULL for torch.linspace(0, 1, n_imp) (det=True).
// Two-pointer sweep over the (monotone) cdf and u.
LAB4D_HD void sample_pdf_ray(const float* __restrict__ b, const float* __restrict__ w, int n_w, int n_imp, float eps, const float* __restrict__ u_in,
                             float* __restrict__ samples, int64_t* __restrict__ inds) {
  // normaliser: torch.sum(weights + eps, -1) (render_utils.py:203-204) in the reference's own (CPU) order of additions
  const float totf = n_w >= 8 ? torch_cpu_row_sum<8>(w, n_w, eps) : torch_cpu_row_sum<1>(w, n_w, eps);
  const float step = 1.0f / (float)(n_imp - 1);
  // cdf[j], j = 0..n_w; cdf[0] = 0; torch.cumsum on the CPU accumulates a float row in f64 and rounds each entry to f32
  int j = 0;              // number of cdf entries consumed that are <= u  (searchsorted right=True)
  double run = 0.0;       // f64 running sum of pdf[0..j-1]
  float c_lo = 0.f;       // cdf[j-1] (for j >= 1)
  float c_hi = 0.f;       // cdf[j]
  for (int k = 0; k < n_imp; ++k) {
    // torch.linspace (CPU): start + step*k for the first half, end - step*(n-1-k) for the second, the latter evaluated with a
    // FUSED multiply-add by its vectorised kernel -- fmaf reproduces it bit for bit for every n (checked for n = 16..128 in
    // tests/test_sample_pdf_host.py::test_linspace_arithmetic); an unfused product does not (n = 16, 64, 128 differ)
    const float u = u_in ? u_in[k] : ((k < n_imp / 2) ? mul_rn(step, (float)k) : fmaf(-step, (float)(n_imp - 1 - k), 1.0f));
    while (j <= n_w && c_hi <= u) {  // advance while cdf[j] <= u
      c_lo = c_hi;
      ++j;
      if (j <= n_w) {
        run += (double)div_rn(add_rn(w[j - 1], eps), totf);
        c_hi = (float)run;
      }
    }
    // inds = j (count of entries <= u); below = max(j-1,0); above = min(j, n_w)
    const int below = j - 1 < 0 ? 0 : j - 1;
    const int above = j > n_w ? n_w : j;
    const float cb = (j == 0) ? c_hi : c_lo;              // cdf[below]
    const float ca = (j > n_w) ? c_lo : c_hi;              // cdf[above]
    float denom = ca - cb;
    if (denom < eps) denom = 1.0f;
    const float b0 = b[below], b1 = b[above];
    samples[k] = add_rn(b0, mul_rn(div_rn(u - cb, denom), b1 - b0));  // separate tensor ops in the reference: unfused
    inds[k] = (int64_t)j;
  }
}

}  // namespace lab4d_pdf
