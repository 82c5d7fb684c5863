// g++ build of lab4d_amd/csrc/ingest_math.hpp (the SAME header the gfx950 kernel compiles) for CPU-side checks -- test infrastructure.
//   g++ -O2 -ffp-contract=off -shared -fPIC -I lab4d_amd/csrc tests/host_harness/ingest_host.cpp -o ingest_host.so
#include "ingest_math.hpp"
using namespace lab4d_ingest;

extern "C" {
// feature (N, FC) fp32 for the pixels xy (N,2) of one frame's feature map (FR, FR, FC) in fp16 (f16 = 1) or fp32
void ingest_host_bilinear(const void* feat, int f16, int FR, int FC, const int* xy, int N, int H, float* out) {
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < FC; ++c)
      out[(long)n * FC + c] = f16 ? bilinear_channel<true>(feat, FR, FC, c, xy[2 * n], xy[2 * n + 1], H)
                                  : bilinear_channel<false>(feat, FR, FC, c, xy[2 * n], xy[2 * n + 1], H);
}
void ingest_host_double_to_half(const double* in, int n, uint16_t* out) {
  for (int i = 0; i < n; ++i) out[i] = double_to_half(in[i]);
}
void ingest_host_half_to_double(const uint16_t* in, int n, double* out) {
  for (int i = 0; i < n; ++i) out[i] = half_to_double(in[i]);
}
}
