// Probe (measurement only): the chain kernels' core loop with TWO waves per SIMD (512-thread workgroups, 256 registers per wave),
// each wave owning a 32-sample tile: per step 16 weight groups read from LDS (ds_read_b128, shared by the 8 waves) feeding 16
// v_mfma_f32_32x32x16_bf16 of ONE 32 x 32 output tile per wave -- per SIMD the same 32 MFMAs = 1024 cycles per step as
// mfma_core.hip's one-wave design, with the same bytes stored / loaded per SIMD and step.
// W0 reads + MFMAs (two accumulation chains, summed by the epilogue); W1 + one workgroup barrier per step; W2 + the weight stash
// (2 ds_write_b128 per wave and step); W3 + a 90-instruction VALU epilogue on the previous tile; W7 = W3 + the tile store (2 x 1 KiB
// streaming stores per wave and step = 16 KiB per CU and step, as V7); W9 = W7 + 2 L2-resident 16-byte loads per wave and step consumed
// one step later (as V9); W10 = W9 with ONE accumulation chain; W11 = W9 with the barrier every SECOND step (two-step weight stash).
// Compare with profiles/r02_mfma_core_probe.json: V1 0.61, V3 0.60, V7 0.44, V9 0.42 of the MFMA peak.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_core2 tools/probes/mfma_core2.hip && /tmp/mfma_core2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int V>
__global__ void __launch_bounds__(512) k_core2(float* out, int steps, char* sbuf, const uint4* table) {
  constexpr bool BAR = (V >= 1), STASH = (V >= 2), EPI = (V >= 3), STORES = (V >= 7), LOADS = (V >= 9);
  constexpr int CHAINS = (V == 10) ? 1 : 2;
  const int wave = blockIdx.x * 8 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __shared__ uint4 abuf[2 * 16 * 64];
  for (int i = threadIdx.x; i < 2 * 16 * 64; i += 512) abuf[i] = make_uint4(0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  __syncthreads();
  uint4 B[16];  // the wave's 32-sample x 256-feature activation tile (B operand), resident
#pragma unroll
  for (int g = 0; g < 16; ++g) B[g] = make_uint4(0x3c003c00u + lane + g, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  f32x16_t acc[2], prev;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; prev[r] = (float)(lane + r); }
  float sink = 0.f;
  u32x4_t wl[2] = {};
  uint4 A[16];
  for (int s0 = 0; s0 < steps; s0 += 4)
#pragma unroll
  for (int sj = 0; sj < 4; ++sj) {
    const int s = s0 + sj;
    if (BAR && (V != 11 || (s & 1) == 0)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int buf = s & 1;
#pragma unroll
    for (int g = 0; g < 16; ++g) A[g] = abuf[(buf * 16 + g) * 64 + lane];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      bf16x8_t av, bv;
      __builtin_memcpy(&av, &A[g], 16);
      __builtin_memcpy(&bv, &B[g], 16);
      const int c = CHAINS == 2 ? (g & 1) : 0;
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[c], 0, 0, 0);
    }
    if (LOADS) {
      sink += __uint_as_float(wl[0].x ^ wl[1].y);
#pragma unroll
      for (int i = 0; i < 2; ++i) wl[i] = ((const __attribute__((address_space(1))) u32x4_t*)table)[(((s * 2 + i) * 61 + wave) & 1023) * 64 + lane];
    }
    if (EPI) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float x = prev[r];
        x = x * 1.0001f + 0.5f; x = fmaxf(x, 0.f); x = x * 0.999f - 0.25f; x = fminf(x, 1e6f); x = x + (float)r;
        prev[r] = x;
      }
      sink += prev[0] + prev[15];
    }
    if (STORES) {
      const int n = lane & 31, h = lane >> 5, q = n & 3, k = n >> 2;
      __attribute__((address_space(1))) char* base = (__attribute__((address_space(1))) char*)sbuf + ((size_t)wave * 4096 + (size_t)(s & 4095)) * 2048;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        u32x4_t v = {__float_as_uint(prev[8 * i]), __float_as_uint(prev[8 * i + 1]), __float_as_uint(prev[8 * i + 2]), __float_as_uint(prev[8 * i + 3])};
        const unsigned off = (unsigned)((8 * i + 4 * h + q) * 128 + 16 * k);
        __builtin_nontemporal_store(v, (__attribute__((address_space(1))) u32x4_t*)(base + off));
      }
    }
    if (STASH) {
#pragma unroll
      for (int i = 0; i < 2; ++i) abuf[((buf ^ 1) * 16 + wid + 8 * i) * 64 + lane] = make_uint4(0x3f803f80u + s, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + i);
    }
    if (EPI) {
#pragma unroll
      for (int r = 0; r < 16; ++r) prev[r] = acc[0][r] + (CHAINS == 2 ? acc[1][r] : 0.f);
    }
  }
  float v = sink;
#pragma unroll
  for (int r = 0; r < 16; ++r) v += acc[0][r] + acc[1][r];
  out[blockIdx.x * 512 + threadIdx.x] = v;
}

template <int V>
void run(float* out, int steps, char* sbuf, const uint4* table) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_core2<V>), dim3(256), dim3(512), 0, 0, out, 64, sbuf, table);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_core2<V>), dim3(256), dim3(512), 0, 0, out, steps, sbuf, table);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us_per_step = ms * 1e3 / steps;
  printf("  \"W%d\": {\"us_per_step\": %.4f, \"cycles_at_2.4GHz\": %.0f, \"mfma_duty\": %.3f},\n", V, us_per_step, us_per_step * 2400.0, 1024.0 / (us_per_step * 2400.0));
}

int main() {
  float* out; CK(hipMalloc(&out, 256 * 512 * 4));
  const int steps = 20000;
  printf("{\n");
  char* sbuf; CK(hipMalloc(&sbuf, (size_t)2048 * 4096 * 2048));  // 8 MiB per wave, 16 GiB: the stores stream to HBM
  uint4* table; CK(hipMalloc(&table, 1 << 20)); CK(hipMemset(table, 1, 1 << 20));
  run<0>(out, steps, sbuf, table); run<1>(out, steps, sbuf, table); run<2>(out, steps, sbuf, table); run<3>(out, steps, sbuf, table);
  run<7>(out, steps, sbuf, table); run<9>(out, steps, sbuf, table); run<10>(out, steps, sbuf, table); run<11>(out, steps, sbuf, table);
  printf("  \"ideal_cycles\": 1024\n}\n");
  return 0;
}
