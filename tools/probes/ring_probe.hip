// Probe (measurement only, not product code; VERDICT r04 "next" 3a): can activations / dZ tiles stream from one CU to another WITHOUT a round trip
// through HBM?  The byte-reducing design for the 256-wide nets is a layer-stationary pipeline: a group of CUs owns a layer (W, W^T
// and dW resident in registers), 64 KiB tiles (128 samples x 256 features bf16) stream CU -> CU.  That only pays if the hand-off runs at more than the
// HBM rate the current design gets (5-6 TB/s aggregate) and its traffic stays in the L2 (4 MiB per XCD) or the Infinity Cache (256 MiB, memory side).
//
// Here: P producer workgroups each write a ring of R slots x 64 KiB, P consumer workgroups read them; hand-off per slot through a pair of sequence
// flags (agent-scope release / acquire: that is what makes a tile written on one CU visible on another -- L1 write-through + invalidate inside an
// XCD, L2 write-back + invalidate across XCDs).  Pairing: SAME XCD (workgroup i with i + 8: the dispatcher deals consecutive workgroup ids round-robin
// over the 8 XCDs) or CROSS XCD (i with i + 1).  Ring bytes in flight = P x R x 64 KiB, swept from 8 MiB to 1 GiB; a plain copy kernel of the same
// bytes (producer and consumer NOT synchronised: write a buffer, then read it in a second launch) gives the HBM reference.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/ring_probe.bin tools/probes/ring_probe.hip && tools/probes/ring_probe.bin [tiles_per_pair]
// Output: one JSON line per configuration: {"pairing", "pairs", "slots", "ring_mib", "gbps", "ok"} (gbps = tile bytes handed over per second; each byte is
// written once and read once).  Under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) the per-kernel counters say how much of it
// crossed the L2 <-> fabric boundary.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

constexpr int TILE_BYTES = 64 * 1024;          // one 128-sample x 256-feature bf16 tile
constexpr int THREADS = 512;                   // the chain kernels' workgroup
constexpr int VEC_PER_THREAD = TILE_BYTES / 16 / THREADS;  // 8 x 16 B per thread and tile
constexpr long SPIN_LIMIT = 1L << 24;

// MODE 0: the memory model's own hand-off (agent-scope release / acquire fences: L2 write-back + invalidate on this part).
// MODE 1..3: NO fences -- the tile's stores / loads carry cache-scope bits instead and the flag follows an s_waitcnt vmcnt(0):
//   1: plain stores (the L1 is write-through: they land in the producer's L2), loads sc1 (agent scope: past the consumer's L1);
//   2: stores sc1, loads sc1;   3: stores sc0 sc1, loads sc0 sc1 (system scope: through the L2s).
// Whether a mode is COHERENT for a pairing is what "ok" reports (every vector of every tile carries its round number).
template <int MODE>
__device__ __forceinline__ void st16(u32x4_t* p, u32x4_t v) {
  if constexpr (MODE == 0 || MODE == 1) *p = v;
  else if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int MODE>
__device__ __forceinline__ u32x4_t ld16(const u32x4_t* p) {
  u32x4_t v;
  if constexpr (MODE == 0) v = *p;
  else if constexpr (MODE == 1 || MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// role: even pair member = producer, odd = consumer.  partner_stride: 8 (same XCD) or 1 (neighbouring XCD).
template <int MODE>
__global__ void __launch_bounds__(THREADS) k_ring(u32x4_t* ring, unsigned* ready, unsigned* freed, int slots, int tiles, int partner_stride, unsigned* err,
                                                  float* sink) {
  // workgroup -> (pair, role).  same-XCD pairing: blocks [16 k, 16 k + 8) are producers of pairs 8 k .. 8 k + 7, blocks [16 k + 8, 16 k + 16) their consumers
  // (ids 8 apart: the same XCD).  cross-XCD pairing: block 2 p is the producer, 2 p + 1 the consumer (neighbouring XCDs).
  const int b = blockIdx.x;
  int pair, role;
  if (partner_stride == 8) { pair = (b / 16) * 8 + (b % 8); role = (b % 16) / 8; }
  else { pair = b / 2; role = b & 1; }
  u32x4_t* my_ring = ring + (size_t)pair * slots * (TILE_BYTES / 16);
  unsigned* my_ready = ready + (size_t)pair * slots;
  unsigned* my_freed = freed + (size_t)pair * slots;
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int t = 0; t < tiles; ++t) {
    const int s = t % slots;
    const unsigned round = (unsigned)(t / slots);
    u32x4_t* tile = my_ring + (size_t)s * (TILE_BYTES / 16);
    if (role == 0) {
      // wait until the consumer has released this slot's previous content
      if (round > 0 && tid == 0) {
        long spins = 0;
        while (__hip_atomic_load(&my_freed[s], MODE == 0 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > SPIN_LIMIT) { atomicExch(err, 1u); break; }
        }
      }
      __syncthreads();
      u32x4_t v = {(unsigned)t, (unsigned)tid, (unsigned)pair, 0x3f800000u};
#pragma unroll
      for (int i = 0; i < VEC_PER_THREAD; ++i) st16<MODE>(&tile[i * THREADS + tid], v);
      if constexpr (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have been acknowledged
      __syncthreads();  // (every wave's stores issued / acknowledged; MODE 0: the release below orders them in front of the flag)
      if (tid == 0) __hip_atomic_store(&my_ready[s], round + 1, MODE == 0 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (tid == 0) {
        long spins = 0;
        while (__hip_atomic_load(&my_ready[s], MODE == 0 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round + 1) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > SPIN_LIMIT) { atomicExch(err, 2u); break; }
        }
      }
      __syncthreads();
      if constexpr (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every wave's L1 must drop what it may hold of the slot's previous round
      u32x4_t vv[VEC_PER_THREAD];
#pragma unroll
      for (int i = 0; i < VEC_PER_THREAD; ++i) vv[i] = ld16<MODE>(&tile[i * THREADS + tid]);
      if constexpr (MODE != 0) {  // the loaded registers are in/out operands of the wait: no use of them can be scheduled in front of it
        static_assert(VEC_PER_THREAD == 8, "operand list below");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vv[4]), "+v"(vv[5]), "+v"(vv[6]), "+v"(vv[7])::"memory");
      }
#pragma unroll
      for (int i = 0; i < VEC_PER_THREAD; ++i) acc += __uint_as_float(vv[i].w) + (vv[i].x == (unsigned)t ? 0.f : 1e9f);  // (wrong round -> visible in the sink)
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&my_freed[s], round + 1, MODE == 0 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (role == 1) sink[(size_t)pair * THREADS + tid] = acc;
}

// HBM reference: the same bytes written by one launch and read by the next (no hand-off, buffer far larger than any cache when `bytes` is)
__global__ void __launch_bounds__(THREADS) k_write(u32x4_t* p, size_t n) {
  u32x4_t v = {1u, 2u, 3u, 0x3f800000u};
  for (size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * THREADS) p[i] = v;
}
__global__ void __launch_bounds__(THREADS) k_read(const u32x4_t* p, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * THREADS) acc += __uint_as_float(p[i].w);
  sink[blockIdx.x * (size_t)THREADS + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 1024;  // per pair: 64 MiB handed over per pair
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  const int pairs = n_cu / 2;  // one workgroup per CU: every producer and every consumer resident at once
  const int slot_sweep[] = {1, 2, 4, 8, 16, 32, 64, 128};
  const int max_slots = 128;
  u32x4_t* ring;
  unsigned *ready, *freed, *err;
  float* sink;
  CK(hipMalloc(&ring, (size_t)pairs * max_slots * TILE_BYTES));
  CK(hipMalloc(&ready, (size_t)pairs * max_slots * 4));
  CK(hipMalloc(&freed, (size_t)pairs * max_slots * 4));
  CK(hipMalloc(&err, 4));
  CK(hipMalloc(&sink, (size_t)n_cu * THREADS * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 4; ++mode)
  for (int stride : {8, 1}) {
    for (int slots : slot_sweep) {
      if (mode > 0 && !(slots == 1 || slots == 4 || slots == 32 || slots == 128)) continue;
      float best = 1e30f;
      unsigned herr = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ready, 0, (size_t)pairs * max_slots * 4));
        CK(hipMemset(freed, 0, (size_t)pairs * max_slots * 4));
        CK(hipMemset(err, 0, 4));
        CK(hipEventRecord(e0));
        auto kern = mode == 0 ? k_ring<0> : mode == 1 ? k_ring<1> : mode == 2 ? k_ring<2> : k_ring<3>;
        hipLaunchKernelGGL(kern, dim3(2 * pairs), dim3(THREADS), 0, 0, ring, ready, freed, slots, tiles, stride, err, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        if (herr) break;
        if (ms < best) best = ms;
      }
      static float hs[512 * 512];
      CK(hipMemcpy(hs, sink, (size_t)pairs * THREADS * 4, hipMemcpyDeviceToHost));
      float s0 = (float)tiles * VEC_PER_THREAD;
      for (int i = 0; i < pairs * THREADS; ++i) if (hs[i] != s0) { s0 = hs[i]; break; }  // any consumer thread that saw a stale / foreign vector
      const double bytes = (double)pairs * tiles * TILE_BYTES;
      printf("{\"mode\": %d, \"pairing\": \"%s\", \"pairs\": %d, \"slots\": %d, \"ring_mib\": %.1f, \"tiles_per_pair\": %d, \"ms\": %.3f, \"gbps\": %.1f, \"ok\": %s}\n",
             mode, stride == 8 ? "same_xcd" : "cross_xcd", pairs, slots, (double)pairs * slots * TILE_BYTES / 1048576.0, tiles, best, bytes / best / 1e6,
             (herr == 0 && s0 == (float)tiles * VEC_PER_THREAD) ? "true" : "false");
      fflush(stdout);
    }
  }
  // HBM reference at 4 GiB
  {
    const size_t bytes = 4ull << 30;
    u32x4_t* big;
    CK(hipMalloc(&big, bytes));
    float msw = 1e30f, msr = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      float ms;
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_write, dim3(n_cu * 4), dim3(THREADS), 0, 0, big, bytes / 16);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < msw) msw = ms;
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_read, dim3(n_cu), dim3(THREADS), 0, 0, big, bytes / 16, sink);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < msr) msr = ms;
    }
    printf("{\"pairing\": \"hbm_reference\", \"bytes_gib\": 4, \"write_gbps\": %.1f, \"read_gbps\": %.1f}\n", bytes / msw / 1e6, bytes / msr / 1e6);
  }
  return 0;
}
