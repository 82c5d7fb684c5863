// CPU build of lab4d_amd/csrc/hashgrid_math.hpp for the test-suite (see fk_host.cpp).  TEST INFRASTRUCTURE ONLY.
#include "hashgrid_math.hpp"

using namespace lab4d_hash;

extern "C" int hashgrid_host_forward(const float* x, const float* table, const int* res, int S, int L, int log2_T, int F, float* out) {
  const size_t slab = ((size_t)1 << log2_T) * F;
  for (long s = 0; s < S; ++s)
    for (int l = 0; l < L; ++l) encode_level(x + 3 * s, table + l * slab, res[l], log2_T, F, out + (size_t)s * L * F + l * F);
  return 0;
}

extern "C" int hashgrid_host_backward(const float* x, const float* table, const int* res, const float* g_out, int S, int L, int log2_T, int F,
                                      float* g_table, float* g_x) {
  const size_t slab = ((size_t)1 << log2_T) * F;
  for (long s = 0; s < S; ++s) {
    float gx[3] = {0.f, 0.f, 0.f};
    for (int l = 0; l < L; ++l)
      encode_level_bwd(x + 3 * s, table + l * slab, res[l], log2_T, F, g_out + (size_t)s * L * F + l * F, g_table ? g_table + l * slab : nullptr, gx);
    if (g_x) { g_x[3 * s] = gx[0]; g_x[3 * s + 1] = gx[1]; g_x[3 * s + 2] = gx[2]; }
  }
  return 0;
}
