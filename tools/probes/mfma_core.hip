// Probe (measurement only): how fast can ONE wave per SIMD run the chain kernels' core loop -- per "step" 16 weight groups read from
// LDS (ds_read_b128, shared by the 4 waves of the workgroup) feeding 32 v_mfma_f32_32x32x16_bf16 (2 n-tiles) -- with nothing else?
// Variants: V0 reads + MFMAs; V1 + one s_barrier per step; V2 + 4 ds_write_b128 per wave and step (the weight stash);
// V3 = V2 + a 180-instruction VALU epilogue on the previous accumulators (independent of the MFMAs in flight);
// V5 = no LDS reads at all (A held in registers): the MFMA rate of ONE wave per SIMD with the kernel's two dependent accumulation chains;
// V6 = V5 with FOUR independent chains (the same 32 MFMAs per step spread over 4 accumulators).
// V4 = V1 with PROGRESSIVE reads: each group's register is re-read for the next step right behind the MFMAs that consumed it.
// V7 = V3 + the tile store of a step (4 x 1 KiB streaming stores per wave, the chain kernels' address pattern, 16 KiB per CU and step);
// V8 = V5 + the same stores; V9 = V7 + 4 L2-resident 16-byte loads per wave and step (weight-fetch stand-in)
// consumed one step later.  profiles/r02_store_ack.json: the same stores cost 45 ns per step next to a SLEEPING wave.
// Store-form variants of V8 for the next round (same 4 KiB per wave and step): V13 data in AGPRs (global_store with an a[] source);
// V14 sixteen 4-byte-per-lane stores; V15 buffer_store_dwordx4 (SGPR resource + 32-bit lane offset); V16 the four stores issued in
// the MIDDLE of the MFMA stream (after group 8) instead of behind it.
// Ideal: 32 MFMAs x 32 cycles = 1024 cycles per step.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_core tools/probes/mfma_core.hip && /tmp/mfma_core
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int V>
__global__ void __launch_bounds__(256) k_core(float* out, int steps, char* sbuf, const uint4* table) {
  constexpr bool REGS = (V == 5 || V == 6 || V == 8 || V >= 10);
  constexpr bool MID = (V == 16);  // V12 = V11 with the store data in one of FOUR dedicated register sets (reused four steps later): is the cost a write-after-read interlock on the store's data registers?          // A operand held in registers (no LDS reads)
  constexpr bool EPI = (V >= 3);    // VALU epilogue stand-in (as recorded in profiles/r02_mfma_core_probe.json: V3..V6 all carry it)
  constexpr bool STASH = (V >= 2);
  constexpr bool BAR = (V >= 1 && V < 10);  // V10 = V8 without the step barrier (waves drift apart); V11 = V8, no barrier, ONE store per step
  constexpr bool STORES = (V >= 7), LOADS = (V == 9);
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  u32x4_t wl[4] = {};
  u32x4_t sd[4] = {};
  __shared__ uint4 abuf[2 * 16 * 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * 16 * 64; i += 256) abuf[i] = make_uint4(0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  __syncthreads();
  uint4 B[2][16];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 16; ++g) B[t][g] = make_uint4(0x3c003c00u + lane + g, 0x3c003c00u, 0x3c003c00u + t, 0x3c003c00u);
  f32x16_t acc[2], prev[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; prev[t][r] = (float)(lane + r); }
  float sink = 0.f;
  uint4 A[16];
  f32x16_t acc4[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc4[t][r] = 0.f;
  for (int s0 = 0; s0 < steps; s0 += 4)
#pragma unroll
  for (int sj = 0; sj < 4; ++sj) {
    const int s = s0 + sj;
    if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int buf = s & 1;
    if ((V != 4 && !REGS) || s == 0) {
#pragma unroll
      for (int g = 0; g < 16; ++g) A[g] = abuf[(buf * 16 + g) * 64 + lane];
    }
    auto do_stores = [&]() {
      const int n = lane & 31, h = lane >> 5, q = n & 3, k = n >> 2;
      __attribute__((address_space(1))) char* base = (__attribute__((address_space(1))) char*)sbuf + ((size_t)wave * 4096 + (size_t)(s & 4095)) * 4096;
#pragma unroll
      for (int i = 0; i < (V == 11 || V == 12 ? 1 : 4); ++i) {
        u32x4_t v = {__float_as_uint(prev[0][4 * i]), __float_as_uint(prev[0][4 * i + 1]), __float_as_uint(prev[1][4 * i + 2]), __float_as_uint(prev[1][4 * i + 3])};
        const unsigned off = (unsigned)((8 * i + 4 * h + q) * 128 + 16 * k);
        if (V == 12) {
          sd[sj] = v;
          asm volatile("" : "+v"(sd[sj]));  // keep the set in its own registers
          v = sd[sj];
        }
        if (V == 13) {  // data from accumulation registers
          asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(base + off), "a"(v) : "memory");
        } else if (V == 14) {  // the same 1 KiB as four 4-byte-per-lane stores (256 B each)
          asm volatile("global_store_dword %0, %1, off nt\n\tglobal_store_dword %0, %2, off offset:4 nt\n\t"
                       "global_store_dword %0, %3, off offset:8 nt\n\tglobal_store_dword %0, %4, off offset:12 nt"
                       ::"v"(base + off), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w) : "memory");
        } else if (V == 15) {  // MUBUF: wave-uniform 128-bit resource in SGPRs, 32-bit lane offset
          char* tile = sbuf + ((size_t)wave * 4096 + (size_t)(s & 4095)) * 4096;
          const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 4096, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)off, 0, 2 /* slc/nt */);
        } else {
          __builtin_nontemporal_store(v, (__attribute__((address_space(1))) u32x4_t*)(base + off));
        }
      }
    };
#pragma unroll
    for (int g = 0; g < 16; ++g) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8_t av, bv;
        __builtin_memcpy(&av, &A[g], 16);
        __builtin_memcpy(&bv, &B[t][g], 16);
        if (V == 6) acc4[2 * t + (g & 1)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc4[2 * t + (g & 1)], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[t], 0, 0, 0);
      }
      if (V == 4) A[g] = abuf[((buf ^ 1) * 16 + g) * 64 + lane];
      if (MID && g == 8) {
        __builtin_amdgcn_sched_barrier(0);  // pin the stores between the MFMAs of groups 8 and 9
        do_stores();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (LOADS) {
      sink += __uint_as_float(wl[0].x ^ wl[1].y ^ wl[2].z ^ wl[3].w);  // consume last step's loads, request this step's
#pragma unroll
      for (int i = 0; i < 4; ++i) wl[i] = ((const __attribute__((address_space(1))) u32x4_t*)table)[(((s * 4 + i) * 61 + wave) & 1023) * 64 + lane];
    }
    if (EPI) {  // stand-in epilogue on the PREVIOUS tile: ~180 dependent-free VALU ops
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float x = prev[t][r];
          x = x * 1.0001f + 0.5f; x = fmaxf(x, 0.f); x = x * 0.999f - 0.25f; x = fminf(x, 1e6f); x = x + (float)r;
          prev[t][r] = x;
        }
      sink += prev[0][0] + prev[1][15];
    }
    if (STORES && !MID) do_stores();
    if (STASH) {
#pragma unroll
      for (int i = 0; i < 4; ++i) abuf[((buf ^ 1) * 16 + wid + 4 * i) * 64 + lane] = make_uint4(0x3f803f80u + s, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + i);
    }
    if (EPI) {
#pragma unroll
      for (int t = 0; t < 2; ++t) prev[t] = acc[t];
    }
  }
  float v = sink;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) v += acc[t][r] + acc4[t][r] + acc4[t + 2][r];
  out[blockIdx.x * 256 + threadIdx.x] = v;
}

template <int V>
void run(float* out, int steps, char* sbuf, const uint4* table) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_core<V>), dim3(256), dim3(256), 0, 0, out, 64, sbuf, table);  // steps: multiples of 4
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_core<V>), dim3(256), dim3(256), 0, 0, out, steps, sbuf, table);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us_per_step = ms * 1e3 / steps;
  printf("  \"V%d\": {\"us_per_step\": %.4f, \"cycles_at_2.4GHz\": %.0f, \"mfma_duty\": %.3f},\n", V, us_per_step, us_per_step * 2400.0, 1024.0 / (us_per_step * 2400.0));
}

int main() {
  float* out; CK(hipMalloc(&out, 256 * 256 * 4));
  const int steps = 20000;
  printf("{\n");
  char* sbuf; CK(hipMalloc(&sbuf, (size_t)1024 * 4096 * 4096));  // 16 MiB per wave, 16 GiB: the stores stream to HBM
  uint4* table; CK(hipMalloc(&table, 1 << 20)); CK(hipMemset(table, 1, 1 << 20));
  run<0>(out, steps, sbuf, table); run<1>(out, steps, sbuf, table); run<2>(out, steps, sbuf, table); run<3>(out, steps, sbuf, table); run<4>(out, steps, sbuf, table);
  run<5>(out, steps, sbuf, table); run<6>(out, steps, sbuf, table); run<7>(out, steps, sbuf, table); run<8>(out, steps, sbuf, table); run<9>(out, steps, sbuf, table); run<10>(out, steps, sbuf, table); run<11>(out, steps, sbuf, table); run<12>(out, steps, sbuf, table);
  run<13>(out, steps, sbuf, table); run<14>(out, steps, sbuf, table); run<15>(out, steps, sbuf, table); run<16>(out, steps, sbuf, table);
  printf("  \"ideal_cycles\": 1024\n}\n");
  return 0;
}
