// Probe (measurement only, not product code): what bounds the hash-grid table gradient -- k_hashgrid_bwd sits at ~25 G fp32 atomic adds per second
// (profiles/r06_hash*.json), 4 % of the HBM roof.  Questions:
//   1. where do device-scope fp32 atomics execute on this 8-XCD part, and how fast are they when the target fits an L2 (1 MiB) vs not (64 MiB)?
//   2. are atomics of a NARROWER scope (workgroup / wavefront: no sc1) faster, and are they still correct when (a) every workgroup may hit every
//      address, (b) the address space is partitioned by the XCD the workgroup really runs on (XCC_ID hardware register), so that all atomics to an
//      address come through ONE L2?
//   3. the packed 2 x bf16 / 2 x f16 atomics (one instruction per F = 2 vertex): rate.
// Every address receives a known number of +1.0 adds (exactly representable): the sum check says whether updates were lost.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/atomic_scope.bin tools/probes/atomic_scope.hip && tools/probes/atomic_scope.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
typedef short v2s __attribute__((ext_vector_type(2)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));

// MODE 0: agent scope fp32; 1: workgroup scope fp32; 2: wavefront scope fp32; 3: packed bf16 (agent); 4: packed f16 (agent)
// PART 0: every workgroup addresses the whole buffer; 1: the eighth of the buffer that belongs to the XCD it runs on
template <int MODE, int PART>
__global__ void __launch_bounds__(256) k_atomics(float* buf, unsigned n_mask, int per_thread, unsigned* xcd_hist) {
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  const unsigned x = xcc_id();
  if (threadIdx.x == 0) atomicAdd(&xcd_hist[x * 2 + ((blockIdx.x & 7u) == x ? 0 : 1)], 1u);  // [xcd][0]: blockIdx % 8 == XCC_ID, [1]: not
  for (int k = 0; k < per_thread; ++k) {
    unsigned a = hash32(gid * 977u + (unsigned)k * 0x9e3779b9u) & n_mask;
    if (PART) a = (a & (n_mask >> 3)) | (x * ((n_mask + 1u) >> 3));
    if constexpr (MODE == 0) __hip_atomic_fetch_add(buf + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (MODE == 1) __hip_atomic_fetch_add(buf + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if constexpr (MODE == 2) __hip_atomic_fetch_add(buf + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else if constexpr (MODE == 3) {
      v2s one = {(short)0x3f80, (short)0x3f80};
      __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) v2s*)(buf + a), one);
    } else {
      v2h one = {(_Float16)1.0f, (_Float16)1.0f};
      __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) v2h*)(buf + a), one);
    }
  }
}

template <int MODE, int PART>
void run(const char* name, float* buf, unsigned n, int blocks, int per_thread, unsigned* hist) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  double sum = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(buf, 0, (size_t)n * 4));
    CK(hipMemset(hist, 0, 64));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_atomics<MODE, PART>), dim3(blocks), dim3(256), 0, 0, buf, n - 1, per_thread, hist);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  std::vector<float> h(n);
  CK(hipMemcpy(h.data(), buf, (size_t)n * 4, hipMemcpyDeviceToHost));
  const double total = (double)blocks * 256 * per_thread;
  if (MODE <= 2) for (unsigned i = 0; i < n; ++i) sum += h[i];
  else if (MODE == 3) for (unsigned i = 0; i < n; ++i) { unsigned w; std::memcpy(&w, &h[i], 4); unsigned lo = (w & 0xffffu) << 16; float f; std::memcpy(&f, &lo, 4); sum += f; }
  else for (unsigned i = 0; i < n; ++i) { unsigned w; std::memcpy(&w, &h[i], 4); _Float16 hh; unsigned short s = (unsigned short)(w & 0xffffu); std::memcpy(&hh, &s, 2); sum += (float)hh; }
  unsigned hh[16];
  CK(hipMemcpy(hh, hist, 64, hipMemcpyDeviceToHost));
  unsigned match = 0, mism = 0;
  for (int i = 0; i < 8; ++i) { match += hh[2 * i]; mism += hh[2 * i + 1]; }
  printf("{\"variant\": \"%s\", \"buffer_mib\": %.1f, \"atomics\": %.0f, \"ms\": %.3f, \"g_atomics_per_s\": %.1f, \"sum_ratio\": %.6f, \"blocks_on_xcd_eq_blockidx_mod8\": %u, \"blocks_elsewhere\": %u}\n",
         name, n * 4.0 / 1048576.0, total, best, total / best / 1e6, sum / total, match, mism);
  fflush(stdout);
}

int main() {
  float* buf;
  unsigned* hist;
  const unsigned n_big = 1u << 24, n_small = 1u << 18;
  CK(hipMalloc(&buf, (size_t)n_big * 4));
  CK(hipMalloc(&hist, 64));
  const int blocks = 4096, per = 64;  // 67 M atomics per launch
  for (unsigned n : {n_small, n_big}) {
    run<0, 0>("agent_fp32_all", buf, n, blocks, per, hist);
    run<1, 0>("workgroup_fp32_all", buf, n, blocks, per, hist);
    run<2, 0>("wavefront_fp32_all", buf, n, blocks, per, hist);
    run<0, 1>("agent_fp32_xcd_partitioned", buf, n, blocks, per, hist);
    run<1, 1>("workgroup_fp32_xcd_partitioned", buf, n, blocks, per, hist);
    run<2, 1>("wavefront_fp32_xcd_partitioned", buf, n, blocks, per, hist);
    run<3, 0>("agent_pk_bf16_all", buf, n, blocks, per, hist);
    run<4, 0>("agent_pk_f16_all", buf, n, blocks, per, hist);
    run<3, 1>("agent_pk_bf16_xcd_partitioned", buf, n, blocks, per, hist);
  }
  return 0;
}
