// Skeleton forward kinematics on gfx950: joint angles (rows, B, 3) -> bone dual quaternions, forward and adjoint.
// SURVEY.md 8f row 1 (the per-frame articulation path in front of the skinning kernels).  Contract: include/lab4d_pose.h.
// The arithmetic lives in fk_math.hpp (shared with the CPU test harness); here: one thread per row, the tree in private
// arrays.  rows = frames of the step (<= a few hundred): this is a launch-count problem, not a bandwidth one -- the
// reference spends ~150 launches forward and ~400 backward on it, this is one launch each way.
#include "common.hpp"
#include "fk_math.hpp"

namespace lab4d {
using namespace lab4d_fk;

// LEN = 0: local joints given per row (fk_se3);  LEN = 1: local joints = rest_local * symmetrised bone length.
template <int LEN>
__global__ void __launch_bounds__(64) k_fk_fwd(const float* __restrict__ so3, const float* __restrict__ local, const float* __restrict__ rest_local,
                                                const float* __restrict__ loglen, const float* __restrict__ logscale,
                                                const float* __restrict__ shift, const int* __restrict__ order, const int* __restrict__ parent,
                                                const int* __restrict__ symm, int R, int B, int bones, float* __restrict__ qr, float* __restrict__ qd) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const Skel sk{B, order, parent, symm};
  float w[MAXB * 3], loc[MAXB * 3], oq[MAXB * 4], od[MAXB * 4], sh[3];
  for (int k = 0; k < 3 * B; ++k) w[k] = so3[(size_t)r * 3 * B + k];
  if (LEN) {
    float ll[MAXB];
    for (int k = 0; k < B; ++k) ll[k] = loglen[(size_t)r * B + k];
    local_joints_fwd(sk, rest_local, ll, logscale[0], loc);
  } else {
    for (int k = 0; k < 3 * B; ++k) loc[k] = local[(size_t)r * 3 * B + k];
  }
  if (shift) { sh[0] = shift[0]; sh[1] = shift[1]; sh[2] = shift[2]; }
  row_forward(sk, w, loc, shift ? sh : nullptr, bones, oq, od);
  for (int k = 0; k < 4 * B; ++k) {
    qr[(size_t)r * 4 * B + k] = oq[k];
    qd[(size_t)r * 4 * B + k] = od[k];
  }
}

template <int LEN>
__global__ void __launch_bounds__(64) k_fk_bwd(const float* __restrict__ so3, const float* __restrict__ local, const float* __restrict__ rest_local,
                                                const float* __restrict__ loglen, const float* __restrict__ logscale,
                                                const float* __restrict__ shift, const int* __restrict__ order, const int* __restrict__ parent,
                                                const int* __restrict__ symm, const float* __restrict__ g_qr, const float* __restrict__ g_qd, int R,
                                                int B, int bones, float* __restrict__ g_so3, float* __restrict__ g_local,
                                                float* __restrict__ g_loglen, float* __restrict__ g_logscale, float* __restrict__ g_shift) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const Skel sk{B, order, parent, symm};
  float w[MAXB * 3], loc[MAXB * 3], gq[MAXB * 4], gd[MAXB * 4], gw[MAXB * 3], gl[MAXB * 3], ll[MAXB], sh[3], gsh[3];
  for (int k = 0; k < 3 * B; ++k) w[k] = so3[(size_t)r * 3 * B + k];
  if (LEN) {
    for (int k = 0; k < B; ++k) ll[k] = loglen[(size_t)r * B + k];
    local_joints_fwd(sk, rest_local, ll, logscale[0], loc);
  } else {
    for (int k = 0; k < 3 * B; ++k) loc[k] = local[(size_t)r * 3 * B + k];
  }
  for (int k = 0; k < 4 * B; ++k) {
    gq[k] = g_qr[(size_t)r * 4 * B + k];
    gd[k] = g_qd[(size_t)r * 4 * B + k];
  }
  if (shift) { sh[0] = shift[0]; sh[1] = shift[1]; sh[2] = shift[2]; }
  row_backward(sk, w, loc, shift ? sh : nullptr, bones, gq, gd, gw, gl, gsh);
  for (int k = 0; k < 3 * B; ++k) g_so3[(size_t)r * 3 * B + k] = gw[k];
  if (LEN) {
    float gll[MAXB], gls;
    local_joints_bwd(sk, rest_local, ll, logscale[0], gl, gll, &gls);
    for (int k = 0; k < B; ++k) g_loglen[(size_t)r * B + k] = gll[k];
    g_logscale[r] = gls;
  } else {
    for (int k = 0; k < 3 * B; ++k) g_local[(size_t)r * 3 * B + k] = gl[k];
  }
  if (g_shift) { g_shift[3 * (size_t)r] = gsh[0]; g_shift[3 * (size_t)r + 1] = gsh[1]; g_shift[3 * (size_t)r + 2] = gsh[2]; }
}

}  // namespace lab4d

using namespace lab4d;

#define FK_COMMON_CHECKS(name)                                                                                         \
  LAB4D_REQUIRE(so3 && order && parent, name ": null pointer");                                                         \
  LAB4D_REQUIRE(B > 0 && B <= lab4d_fk::MAXB && R >= 0, name ": bad sizes R=%d B=%d (B <= %d)", R, B, lab4d_fk::MAXB); \
  if (R == 0) return LAB4D_OK;

extern "C" int lab4d_fk_forward(const float* so3, const float* local, const float* shift, const int32_t* order, const int32_t* parent, int R, int B,
                                int bones, float* qr, float* qd, void* stream) {
  FK_COMMON_CHECKS("fk_forward");
  LAB4D_REQUIRE(local && qr && qd, "fk_forward: null pointer");
  hipLaunchKernelGGL(k_fk_fwd<0>, dim3(div_up(R, 64)), dim3(64), 0, (hipStream_t)stream, so3, local, nullptr, nullptr, nullptr, shift, order, parent,
                     nullptr, R, B, bones, qr, qd);
  return check_launch("fk_forward");
}

extern "C" int lab4d_fk_backward(const float* so3, const float* local, const float* shift, const int32_t* order, const int32_t* parent,
                                 const float* g_qr, const float* g_qd, int R, int B, int bones, float* g_so3, float* g_local, float* g_shift,
                                 void* stream) {
  FK_COMMON_CHECKS("fk_backward");
  LAB4D_REQUIRE(local && g_qr && g_qd && g_so3 && g_local, "fk_backward: null pointer");
  hipLaunchKernelGGL(k_fk_bwd<0>, dim3(div_up(R, 64)), dim3(64), 0, (hipStream_t)stream, so3, local, nullptr, nullptr, nullptr, shift, order, parent,
                     nullptr, g_qr, g_qd, R, B, bones, g_so3, g_local, nullptr, nullptr, g_shift);
  return check_launch("fk_backward");
}

extern "C" int lab4d_skel_bones_forward(const float* so3, const float* loglen, const float* logscale, const float* rest_local, const float* shift,
                                        const int32_t* order, const int32_t* parent, const int32_t* symm, int R, int B, float* qr, float* qd,
                                        void* stream) {
  FK_COMMON_CHECKS("skel_bones_forward");
  LAB4D_REQUIRE(loglen && logscale && rest_local && symm && qr && qd, "skel_bones_forward: null pointer");
  hipLaunchKernelGGL(k_fk_fwd<1>, dim3(div_up(R, 64)), dim3(64), 0, (hipStream_t)stream, so3, nullptr, rest_local, loglen, logscale, shift, order,
                     parent, symm, R, B, 1, qr, qd);
  return check_launch("skel_bones_forward");
}

extern "C" int lab4d_skel_bones_backward(const float* so3, const float* loglen, const float* logscale, const float* rest_local, const float* shift,
                                         const int32_t* order, const int32_t* parent, const int32_t* symm, const float* g_qr, const float* g_qd,
                                         int R, int B, float* g_so3, float* g_loglen, float* g_logscale, float* g_shift, void* stream) {
  FK_COMMON_CHECKS("skel_bones_backward");
  LAB4D_REQUIRE(loglen && logscale && rest_local && symm && g_qr && g_qd && g_so3 && g_loglen && g_logscale, "skel_bones_backward: null pointer");
  hipLaunchKernelGGL(k_fk_bwd<1>, dim3(div_up(R, 64)), dim3(64), 0, (hipStream_t)stream, so3, nullptr, rest_local, loglen, logscale, shift, order,
                     parent, symm, g_qr, g_qd, R, B, 1, g_so3, nullptr, g_loglen, g_logscale, g_shift);
  return check_launch("skel_bones_backward");
}
