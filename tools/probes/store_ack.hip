// Probe (measurement only): does a per-step LOAD wait pay for the STORES issued in front of it?  gfx9-family vmcnt retires loads and
// stores in issue order, so waiting for a load also waits for every older store.  One wave per SIMD (256 x 4 waves), each step:
//   request one 16-byte-per-lane load from an L2-resident table, issue the chain kernels' tile store (4 x 1 KiB streaming stores,
//   k_mlp_fwd<FgBase> address pattern), "work" for W cycles (s_sleep), and consume the load requested D steps earlier.
// Reported: ns per step for W in {1000, 2000, 3000, 4000} cycles and
//   stores only / loads only / loads + stores with D = 1, 2, 3.   If "loads + stores, D = 1" is slower than both "stores only" and
//   "loads only" at the same W, the load wait is paying for store acknowledgements; larger D shows how old a store has to be.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_ack tools/probes/store_ack.hip && /tmp/store_ack
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NL = 9, F = 256;
constexpr size_t BLK = (size_t)F * 64 + 128;  // elements (bf16) between 64-sample blocks

struct Bufs {
  unsigned short* p[NL];
};
template <bool ST, bool LD, int D, int SLEEP>
__global__ void __launch_bounds__(256) k_steps(Bufs bufs, const u32x4_t* table, int nblocks, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = blockIdx.x * 4 + wid, nwaves = gridDim.x * 4;
  const int n = lane & 31, h = lane >> 5, q = n & 3, k = n >> 2;
  u32x4_t v = {(unsigned)lane, 1u, 2u, 3u};
  u32x4_t ring[D];
#pragma unroll
  for (int d = 0; d < D; ++d) ring[d] = u32x4_t{0u, 0u, 0u, 0u};
  unsigned acc = 0;
  int step = 0;
  for (int b = wave; b < nblocks; b += nwaves) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {  // unrolled: the layer pointers are kernel arguments (scalar loads, off vmcnt)
      GLOBAL_AS char* base = (GLOBAL_AS char*)bufs.p[l] + (size_t)b * BLK * 2;
#pragma unroll 1
      for (int mt0 = 0; mt0 < 8; mt0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {  // D steps per trip: the ring slot of a step is static
          const int mt = mt0 + d;
          if (mt < 8) {
            if (LD) {
              // consume the load requested D steps ago: the explicit counted wait pins the place (younger operations allowed in
              // flight: D-1 loads and, with stores, 4 D stores), exactly what the chain kernels' per-step waits look like
              constexpr int YOUNGER = (D - 1) + (ST ? 4 * D : 0);
              asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ring[d]) : "n"(YOUNGER) : "memory");
              acc += ring[d].x ^ ring[d].w;
              ring[d] = ((const GLOBAL_AS u32x4_t*)table)[((step * 67 + wave) & 1023) * 64 + lane];  // 1 MiB table, L2-resident
            }
            if (ST) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                GLOBAL_AS u32x4_t* p = (GLOBAL_AS u32x4_t*)(base + (size_t)(32 * mt + 8 * i + 4 * h + q) * 128 + 16 * k);
                __builtin_nontemporal_store(v, p);
              }
            }
            __builtin_amdgcn_s_sleep(SLEEP);  // 64 * SLEEP cycles of "matrix work"
            ++step;
          }
        }
      }
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) acc += ring[d].y;
  if (acc == 0x12345678u) *sink = acc;
}

template <class Fn>
float time_ms(Fn&& f, int reps = 3) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const int S = 2097152, nblocks = S / 64;  // 32 blocks per wave x 72 steps = 2304 steps per wave
  const size_t per = (size_t)nblocks * BLK * 2;
  unsigned short* h[NL];
  for (int l = 0; l < NL; ++l) CK(hipMalloc(&h[l], per));
  Bufs d;
  for (int l = 0; l < NL; ++l) d.p[l] = h[l];
  u32x4_t* table; CK(hipMalloc(&table, 1 << 20)); CK(hipMemset(table, 1, 1 << 20));
  unsigned* sink; CK(hipMalloc(&sink, 4));
  const double steps_per_wave = (double)nblocks / 1024.0 * NL * 8;
  printf("{\"steps_per_wave\": %.0f, \"store_bytes_per_step_per_cu\": 16384", steps_per_wave);
#define RUN(name, ST, LD, DD, SL)                                                                                                   \
  { float ms = time_ms([&] { hipLaunchKernelGGL((k_steps<ST, LD, DD, SL>), dim3(256), dim3(256), 0, 0, d, table, nblocks, sink); }); \
    printf(", \"%s_w%d\": %.0f", name, 64 * SL, ms * 1e6 / steps_per_wave); }
#define SWEEP(SL)                              \
  RUN("none", false, false, 1, SL)             \
  RUN("stores", true, false, 1, SL)            \
  RUN("loads_d1", false, true, 1, SL)          \
  RUN("loads_stores_d1", true, true, 1, SL)    \
  RUN("loads_stores_d2", true, true, 2, SL)    \
  RUN("loads_stores_d4", true, true, 4, SL)
  SWEEP(16) SWEEP(32) SWEEP(48) SWEEP(64)
  printf(", \"unit\": \"ns per step\"}\n");
  return 0;
}
