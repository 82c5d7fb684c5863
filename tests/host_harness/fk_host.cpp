// CPU build of lab4d_amd/csrc/fk_math.hpp for the test-suite (g++ -shared): the same row arithmetic the gfx950 kernels of
// fk.hip execute, callable from ctypes, so the `-m "not gpu"` tests can hold it to the oracle and to the reference fixture.
// TEST INFRASTRUCTURE ONLY: never loaded by lab4d_amd/.
#include "fk_math.hpp"

using namespace lab4d_fk;

extern "C" int fk_host_forward(const float* so3, const float* local, const float* shift, const int* order, const int* parent, int R, int B,
                               int bones, float* qr, float* qd) {
  const Skel sk{B, order, parent, nullptr};
  for (int r = 0; r < R; ++r)
    row_forward(sk, so3 + (long)r * 3 * B, local + (long)r * 3 * B, shift, bones, qr + (long)r * 4 * B, qd + (long)r * 4 * B);
  return 0;
}

extern "C" int fk_host_backward(const float* so3, const float* local, const float* shift, const int* order, const int* parent, const float* g_qr,
                                const float* g_qd, int R, int B, int bones, float* g_so3, float* g_local, float* g_shift) {
  const Skel sk{B, order, parent, nullptr};
  for (int r = 0; r < R; ++r)
    row_backward(sk, so3 + (long)r * 3 * B, local + (long)r * 3 * B, shift, bones, g_qr + (long)r * 4 * B, g_qd + (long)r * 4 * B,
                 g_so3 + (long)r * 3 * B, g_local + (long)r * 3 * B, g_shift + 3 * (long)r);
  return 0;
}

extern "C" int skel_host_forward(const float* so3, const float* loglen, float logscale, const float* rest_local, const float* shift, const int* order,
                                 const int* parent, const int* symm, int R, int B, float* qr, float* qd) {
  const Skel sk{B, order, parent, symm};
  float loc[MAXB * 3];
  for (int r = 0; r < R; ++r) {
    local_joints_fwd(sk, rest_local, loglen + (long)r * B, logscale, loc);
    row_forward(sk, so3 + (long)r * 3 * B, loc, shift, 1, qr + (long)r * 4 * B, qd + (long)r * 4 * B);
  }
  return 0;
}

extern "C" int skel_host_backward(const float* so3, const float* loglen, float logscale, const float* rest_local, const float* shift, const int* order,
                                  const int* parent, const int* symm, const float* g_qr, const float* g_qd, int R, int B, float* g_so3,
                                  float* g_loglen, float* g_logscale, float* g_shift) {
  const Skel sk{B, order, parent, symm};
  float loc[MAXB * 3], gl[MAXB * 3];
  for (int r = 0; r < R; ++r) {
    local_joints_fwd(sk, rest_local, loglen + (long)r * B, logscale, loc);
    row_backward(sk, so3 + (long)r * 3 * B, loc, shift, 1, g_qr + (long)r * 4 * B, g_qd + (long)r * 4 * B, g_so3 + (long)r * 3 * B, gl,
                 g_shift + 3 * (long)r);
    local_joints_bwd(sk, rest_local, loglen + (long)r * B, logscale, gl, g_loglen + (long)r * B, g_logscale + r);
  }
  return 0;
}
