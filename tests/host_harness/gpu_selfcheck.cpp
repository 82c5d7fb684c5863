// Native self-check of the hash-grid kernels on the device (no Python): runs lab4d_hashgrid_forward / _backward through the
// C ABI of liblab4d_hip.so and compares with the host build of the same arithmetic (hashgrid_math.hpp).  Exit code 0 = match.
// TEST INFRASTRUCTURE ONLY.   hipcc --offload-arch=gfx950 gpu_selfcheck.cpp -I include -I lab4d_amd/csrc -L lab4d_amd -llab4d_hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "hashgrid_math.hpp"
#include "lab4d_hip.h"

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } \
  } while (0)

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; }

int main() {
  const int S = 20011, L = 8, F = 2, log2_T = 12;
  const int res_h[L] = {4, 6, 10, 16, 25, 40, 64, 101};
  const size_t T = (size_t)1 << log2_T, nt = L * T * F;
  unsigned seed = 7;
  std::vector<float> x(3 * S), tab(nt), g(S * L * F), out(S * L * F), gt(nt, 0.f), gx(3 * S);
  for (auto& v : x) v = frand(seed);
  x[0] = x[1] = x[2] = 0.f; x[3] = x[4] = x[5] = 1.f;
  for (auto& v : tab) v = frand(seed) - 0.5f;
  for (auto& v : g) v = frand(seed) - 0.5f;
  for (int s = 0; s < S; ++s) {
    float a[3] = {0.f, 0.f, 0.f};
    for (int l = 0; l < L; ++l) {
      lab4d_hash::encode_level(&x[3 * s], &tab[l * T * F], res_h[l], log2_T, F, &out[(size_t)s * L * F + l * F]);
      lab4d_hash::encode_level_bwd(&x[3 * s], &tab[l * T * F], res_h[l], log2_T, F, &g[(size_t)s * L * F + l * F], &gt[l * T * F], a);
    }
    gx[3 * s] = a[0]; gx[3 * s + 1] = a[1]; gx[3 * s + 2] = a[2];
  }
  float *dx, *dt, *dg, *dout, *dgt, *dgx;
  int* dres;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dt, nt * 4)); CK(hipMalloc(&dg, g.size() * 4)); CK(hipMalloc(&dout, out.size() * 4));
  CK(hipMalloc(&dgt, nt * 4)); CK(hipMalloc(&dgx, gx.size() * 4)); CK(hipMalloc(&dres, L * 4));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, tab.data(), nt * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dres, res_h, L * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dgt, 0, nt * 4));
  if (lab4d_hashgrid_forward(dx, dt, dres, S, L, log2_T, F, dout, nullptr)) { printf("fwd: %s\n", lab4d_last_error()); return 3; }
  if (lab4d_hashgrid_backward(dx, dt, dres, dg, S, L, log2_T, F, dgt, dgx, nullptr)) { printf("bwd: %s\n", lab4d_last_error()); return 3; }
  CK(hipDeviceSynchronize());
  std::vector<float> o2(out.size()), gt2(nt), gx2(gx.size());
  CK(hipMemcpy(o2.data(), dout, out.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gt2.data(), dgt, nt * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(gx2.data(), dgx, gx.size() * 4, hipMemcpyDeviceToHost));
  float e_out = 0, e_gt = 0, e_gx = 0, m_gt = 0, m_gx = 0;
  for (size_t i = 0; i < out.size(); ++i) e_out = fmaxf(e_out, fabsf(o2[i] - out[i]));
  for (size_t i = 0; i < nt; ++i) { e_gt = fmaxf(e_gt, fabsf(gt2[i] - gt[i])); m_gt = fmaxf(m_gt, fabsf(gt[i])); }
  for (size_t i = 0; i < gx.size(); ++i) { e_gx = fmaxf(e_gx, fabsf(gx2[i] - gx[i])); m_gx = fmaxf(m_gx, fabsf(gx[i])); }
  const bool ok = e_out < 1e-5f && e_gt < 1e-4f * fmaxf(1.f, m_gt) && e_gx < 1e-4f * fmaxf(1.f, m_gx);
  printf("hashgrid selfcheck S=%d L=%d F=%d T=2^%d: max|d out|=%.2e  max|d g_table|=%.2e (of %.2e)  max|d g_x|=%.2e (of %.2e)  -> %s\n", S, L, F, log2_T,
         e_out, e_gt, m_gt, e_gx, m_gx, ok ? "OK" : "MISMATCH");
  return ok ? 0 : 1;
}
