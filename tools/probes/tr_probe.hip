// Probe (measurement only): semantics of ds_read_b64_tr_b16 on gfx950 with ARBITRARY per-lane addresses.
// Hypothesis (generalising the guide's formula for consecutive chunks): inside each 16-lane group, lane i's result element j
// (j = 0..3, 16 bits each) is element (i & 3) of the 8-byte chunk addressed by lane (i >> 2) + 4 j of the same group.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/tr_probe tools/probes/tr_probe.hip && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k(const unsigned* addr_in, unsigned long long* out) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;  // value = element index
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds;
  const unsigned a = base + addr_in[threadIdx.x];  // byte address, 8-byte aligned
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x] = v;
}

int main() {
  std::vector<unsigned> addr(64);
  srand(7);
  // an arbitrary assignment of distinct 8-byte chunks to lanes
  std::vector<int> chunks(1024);
  for (int i = 0; i < 1024; ++i) chunks[i] = i;
  for (int i = 0; i < 64; ++i) { int j = i + rand() % (1024 - i); std::swap(chunks[i], chunks[j]); addr[i] = 8u * (unsigned)chunks[i]; }
  unsigned* da; unsigned long long* dout;
  CK(hipMalloc(&da, 256)); CK(hipMalloc(&dout, 512));
  CK(hipMemcpy(da, addr.data(), 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> out(64);
  CK(hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int g = l & ~15, i = l & 15;
    for (int j = 0; j < 4; ++j) {
      const unsigned got = (unsigned)((out[l] >> (16 * j)) & 0xffff);
      const int src_lane = g + (i >> 2) + 4 * j;
      const unsigned want = addr[src_lane] / 2 + (i & 3);
      if (got != want) { if (bad < 12) printf("lane %d elem %d: got %u want %u\n", l, j, got, want); ++bad; }
    }
  }
  printf("{\"hypothesis\": \"out[lane i of a 16-lane group][j] = chunk(lane (i>>2)+4j)[i&3]\", \"mismatches\": %d}\n", bad);
  if (bad) for (int l = 0; l < 20; ++l) printf("lane %2d addr/2=%4u : %4u %4u %4u %4u\n", l, addr[l] / 2, (unsigned)(out[l] & 0xffff), (unsigned)((out[l] >> 16) & 0xffff), (unsigned)((out[l] >> 32) & 0xffff), (unsigned)((out[l] >> 48) & 0xffff));
  return 0;
}
