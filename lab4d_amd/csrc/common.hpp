// Shared helpers for the gfx950 kernels of liblab4d_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "lab4d_hip.h"

namespace lab4d {

void set_error(const char* fmt, ...);

#define LAB4D_REQUIRE(cond, ...)              \
  do {                                        \
    if (!(cond)) {                            \
      lab4d::set_error(__VA_ARGS__);          \
      return LAB4D_EINVAL;                    \
    }                                         \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return LAB4D_ELAUNCH;
  }
  return LAB4D_OK;
}

// Zero `bytes` (a multiple of 4) at p, ordered on the stream like any kernel.  A fill KERNEL, not hipMemsetAsync: a memset node captured
// into a hipGraph is not ordered against the kernel nodes around it on this runtime (ROCm 7.2) -- on the first replay the accumulator it
// should clear is fresh (zero) memory and everything looks right, from the second replay on the atomics land on the previous replay's
// sums (found in round 3: the loss vector and the per-frame gradient accumulators of a replayed training chunk were garbage).
int zero_async(void* p, size_t bytes, hipStream_t stream);

constexpr int kWave = 64;  // CDNA wavefront

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ float wave_scan_incl(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}
// inclusive suffix sum (sum over lanes >= this lane)
__device__ __forceinline__ float wave_rscan_incl(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_down(v, o, 64);
    if (lane + o < 64) v += t;
  }
  return v;
}

// A rounded fp32 product / sum that the optimiser cannot fuse into an fma.  HIP compiles with -ffp-contract=fast and the
// backend contracts even __fmul_rn / __fadd_rn (found on the first hardware run of the hash grid, round 2), so wherever the
// reference's arithmetic rounds a product before using it, the product is made opaque.
__device__ __forceinline__ float mul_rn(float a, float b) {
  float p = a * b;
  asm volatile("" : "+v"(p));
  return p;
}
__device__ __forceinline__ float add_rn(float a, float b) {
  float p = a + b;
  asm volatile("" : "+v"(p));
  return p;
}

inline int div_up(long a, long b) { return (int)((a + b - 1) / b); }
__device__ __forceinline__ int div_up_dev(int a, int b) { return (a + b - 1) / b; }

}  // namespace lab4d
