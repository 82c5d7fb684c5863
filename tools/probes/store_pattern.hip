// Probe (measurement only): what a 16-byte-per-lane global store costs a CU that is busy with MFMAs, by LANE -> ADDRESS pattern.
// Round 4 finding behind it: in the weights-stationary chain kernels the tile stores are ADDITIVE to the matrix work (4.64 ms without the store
// instructions, 6.08 ms with them, the same with the stores aimed at an L2-resident window) -- ~37 CU cycles per store instruction, so it is the
// store's way through the CU (address / data transfer, coalescing in the texture-addresser), not HBM.
// 8 waves per workgroup, one workgroup per CU; per "layer" a wave issues 64 v_mfma_f32_32x32x16_bf16 and 10 stores of 1 KiB spread between them.
//   P0 no stores            P1 tile pattern of tr_store: the 4 lanes of a quad write 16 B each to 4 DIFFERENT 128-byte rows
//   P2 quad-contiguous: the 4 lanes of a quad write 64 contiguous bytes of ONE row      P3 lane-linear: lane L writes bytes 16 L .. 16 L + 15
// each to a large buffer (HBM) and to a 2 MiB window (L2-resident).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/store_pattern.bin tools/probes/store_pattern.hip && tools/probes/store_pattern.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define GLOBAL_AS __attribute__((address_space(1)))

template <int P>
__global__ void __launch_bounds__(512) k_probe(char* buf, size_t window_mask, int layers, float* out) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3c00 + lane + i); b[i] = (short)(0x3c00 + wid); }
  f32x16_t acc0, acc1;
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  // per-lane byte offset inside a 1 KiB piece (8 rows x 128 B)
  unsigned lo;
  {
    const int i = lane & 15, G = lane >> 4, h = G & 1, S = G >> 1;
    if (P == 1) lo = (unsigned)((4 * h + (i & 3)) * 128 + (32 * S + 8 * (i >> 2)) * 2);
    else if (P == 2) lo = (unsigned)((4 * h + (i >> 2)) * 128 + (32 * S + 8 * (i & 3)) * 2);
    else lo = (unsigned)(lane * 16);
  }
  size_t pos = ((size_t)blockIdx.x * 8 + wid) * 4096;
  const size_t stride = (size_t)gridDim.x * 8 * 4096;
  GLOBAL_AS char* gb = (GLOBAL_AS char*)buf;
  for (int l = 0; l < layers; ++l) {
#pragma unroll
    for (int g = 0; g < 32; ++g) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      if (P != 0 && (g % 3) == 1 && g < 30) {  // 10 stores per layer
        const u32x4_t v = {(unsigned)l, (unsigned)g, (unsigned)lane, 0u};
        const size_t off = (pos + (size_t)(g / 3) * 1024) & window_mask;
        __builtin_nontemporal_store(v, (GLOBAL_AS u32x4_t*)(gb + off + lo));
      }
    }
    pos += stride;
  }
  if (acc0[0] + acc1[1] == 12345.f) out[0] = acc0[3];
}

template <int P>
void run(const char* what, char* buf, size_t bytes, size_t window, float* out) {
  const int layers = 2048, grid = 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t mask = window ? window - 1 : ~(size_t)0;
  hipLaunchKernelGGL(k_probe<P>, dim3(grid), dim3(512), 0, 0, buf, mask, 32, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_probe<P>, dim3(grid), dim3(512), 0, 0, buf, mask, layers, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double ns_per_layer = ms * 1e6 / layers;
  const double tf = 512.0 * 32768.0 * grid * layers / (ms * 1e-3) / 1e12;
  const double gbs = P ? 80.0 * 1024 * grid * layers / (ms * 1e-3) / 1e9 : 0.0;
  printf("{\"pattern\": \"P%d\", \"what\": \"%s\", \"target\": \"%s\", \"ns_per_layer\": %.1f, \"cycles_per_layer_at_2.4GHz\": %.0f, \"frac_of_bf16_mfma_peak\": %.3f, \"store_GB_s\": %.0f}\n", P, what,
         window ? "2 MiB window (L2)" : "HBM", ns_per_layer, ns_per_layer * 2.4, tf / 2500.0, gbs);
}

int main() {
  const size_t bytes = (size_t)256 * 8 * 4096 * 2048 + (1 << 20);  // 16 GiB + slack: every piece of every layer has its own address
  char* buf; float* out;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&out, 1024));
  run<0>("no stores", buf, bytes, 0, out);
  for (size_t window : {(size_t)0, (size_t)(2 << 20)}) {
    run<1>("tile pattern of tr_store: a quad writes 4 x 16 B to 4 rows", buf, bytes, window, out);
    run<2>("quad-contiguous: a quad writes 64 B of one row", buf, bytes, window, out);
    run<3>("lane-linear", buf, bytes, window, out);
  }
  return 0;
}
