// Probe (measurement only, not product code): what write bandwidth can the chain kernels' store pattern reach on MI355X with no MFMA
// work at all?  Pattern = k_mlp_fwd<FgBase> training mode: a wave owns 64-sample blocks; per block and layer it writes 8 M-tiles of
// 4 KiB (4 x 16-byte-per-lane stores of 1 KiB each) into that layer's [64-block][256 features][64] bf16 buffer (block stride
// 256*64+128 elements), 9 layers.  Variants: streaming (nt) vs plain stores; 4 / 8 / 16 waves per CU; plus a plain float4 copy and a
// read-only sweep of the same bytes for reference.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_roof tools/probes/store_roof.hip && /tmp/store_roof
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NL = 9, F = 256;
constexpr size_t BLK = (size_t)F * 64 + 128;  // elements (bf16) between blocks

template <bool NT_, int STEP_WORK>
__global__ void __launch_bounds__(256) k_store(unsigned short* const* bufs, int nblocks, int wpb) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = blockIdx.x * wpb + wid, nwaves = gridDim.x * wpb;
  const int n = lane & 31, h = lane >> 5, q = n & 3, k = n >> 2;
  u32x4_t v = {(unsigned)lane, 1u, 2u, 3u};
  for (int b = wave; b < nblocks; b += nwaves) {
    for (int l = 0; l < NL; ++l) {
      GLOBAL_AS char* base = (GLOBAL_AS char*)bufs[l] + (size_t)b * BLK * 2;
      for (int mt = 0; mt < 8; ++mt) {
        if (STEP_WORK > 0) {  // stand-in for the MFMA time of a step: dependent VALU chain
          unsigned x = v.x;
#pragma unroll 1
          for (int i = 0; i < STEP_WORK; ++i) x = x * 1664525u + 1013904223u;
          v.x = x;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          GLOBAL_AS u32x4_t* p = (GLOBAL_AS u32x4_t*)(base + (size_t)(32 * mt + 8 * i + 4 * h + q) * 128 + 16 * k);
          if (NT_) __builtin_nontemporal_store(v, p);
          else *p = v;
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ a, float* out, size_t n) {
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) *out = s;
}
__global__ void __launch_bounds__(256) k_fill(float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <class Fn>
float time_ms(Fn&& f, int reps = 5) {
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
  const int S = 4194304, nblocks = S / 64;
  const size_t per = (size_t)nblocks * BLK * 2;
  unsigned short* h[NL];
  for (int l = 0; l < NL; ++l) CK(hipMalloc(&h[l], per));
  unsigned short** d; CK(hipMalloc(&d, sizeof(h))); CK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
  const double bytes = (double)S * 512 * NL;
  printf("{\"bytes_per_pass\": %.0f", bytes);
#define RUN(name, NTF, WORK, wpb, grid)                                                                      \
  { float ms = time_ms([&] { hipLaunchKernelGGL((k_store<NTF, WORK>), dim3(grid), dim3(64 * wpb), 0, 0, d, nblocks, wpb); }); \
    printf(", \"%s\": {\"ms\": %.3f, \"TBps\": %.3f}", name, ms, bytes / ms / 1e9); }
  RUN("nt_4waves_per_cu", true, 0, 4, 256)
  RUN("plain_4waves_per_cu", false, 0, 4, 256)
  RUN("nt_8waves_per_cu", true, 0, 4, 512)
  RUN("nt_16waves_per_cu", true, 0, 4, 1024)
  RUN("plain_16waves_per_cu", false, 0, 4, 1024)
  RUN("nt_32waves_per_cu", true, 0, 4, 2048)
  RUN("nt_4waves_per_cu_work300", true, 300, 4, 256)
  RUN("nt_8waves_per_cu_work300", true, 300, 4, 512)
  {
    const size_t n = (size_t)4 << 30;  // 4 GiB each way
    float4 *a, *b; CK(hipMalloc(&a, n)); CK(hipMalloc(&b, n)); float* o; CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 1, n));
    float ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, a, b, n / 16); });
    printf(", \"float4_copy\": {\"ms\": %.3f, \"TBps_rw\": %.3f}", ms, 2.0 * n / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(256 * 8), dim3(256), 0, 0, a, o, n / 16); });
    printf(", \"float4_read\": {\"ms\": %.3f, \"TBps\": %.3f}", ms, 1.0 * n / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_fill, dim3(256 * 8), dim3(256), 0, 0, b, n / 16); });
    printf(", \"float4_fill\": {\"ms\": %.3f, \"TBps\": %.3f}", ms, 1.0 * n / ms / 1e9);
    float msm = time_ms([&] { CK(hipMemsetAsync(b, 0, n, 0)); });
    printf(", \"hipMemset\": {\"ms\": %.3f, \"TBps\": %.3f}", msm, 1.0 * n / msm / 1e9);
  }
  printf("}\n");
  return 0;
}
