// Test probe (NOT part of liblab4d_hip.so): one-wave MFMA tiles with the operand / accumulator lane layouts that csrc/mlp_kernels.hpp
// assumes.  tests/test_gpu_ops.py::test_mfma_layout_probe multiplies asymmetric matrices through it so a wrong layout assumption
// is caught in isolation.  Built by __graft_entry__.build() into tests/host_harness/_build/libmfma_layout.so.
#include <hip/hip_runtime.h>

namespace lab4d {
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ unsigned short f2bf(float x) {  // round-to-nearest-even
  unsigned int u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// A (32 x K) row-major, B (K x 32) row-major, D (32 x 32) row-major.  K = 16 (bf16) or 2 (f32).
__global__ void k_probe_bf16(const float* A, const float* B, float* Dm) {
  const int l = threadIdx.x, i = l & 31, h = l >> 5;
  bf16x8_t a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (short)f2bf(A[i * 16 + 8 * h + j]);      // A[i][k = 8h + j]
    b[j] = (short)f2bf(B[(8 * h + j) * 32 + i]);    // B[k = 8h + j][n = i]
  }
  f32x16_t c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) Dm[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];  // row, col = lane&31
}
__global__ void k_probe_f32(const float* A, const float* B, float* Dm) {
  const int l = threadIdx.x, i = l & 31, h = l >> 5;
  const float a = A[i * 2 + h];      // A[i][k = h]
  const float b = B[h * 32 + i];     // B[k = h][n = i]
  f32x16_t c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) Dm[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];
}
}  // namespace lab4d

extern "C" int mfma_layout_probe(const float* A, const float* B, float* D, int use_bf16, void* stream) {
  if (use_bf16) hipLaunchKernelGGL(lab4d::k_probe_bf16, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D);
  else hipLaunchKernelGGL(lab4d::k_probe_f32, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
