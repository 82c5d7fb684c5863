// Per-sample epilogues of the training query that the reference writes as chains of element-wise tensor operations over
// (M,N,D,3) tensors (contract: include/lab4d_hip.h section 4b):
//   * optical-flow proposal -- NeRF.compute_flow (nnutils/nerf.py:948-997): field_to_cam with the pair partner's pose
//     (nerf.py:846-863), pinhole_projection (utils/geom_utils.py:14-27), difference to the pixel, validity test;
//   * cycle distance -- Deformable.cycle_loss (nnutils/deformable.py:173-198): | forward-warped point - time-t point |;
//   * VolSDF density -- NeRF.forward (nerf.py:186-192).
// ~12 + 4 + 6 + 3 launches forward and twice that backward in the reference (and in round 1 of this repository), each moving
// a 50-150 MB tensor; here one kernel each way: 40 B read + 16 B written per sample forward.  HBM-bound.
#include "common.hpp"

namespace lab4d {

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* p, F3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// quaternion_translation_apply without normalising q (quat_transform.py:416-428 via quaternion_apply):
// (w^2 - v.v) x + 2 (v.x) v + 2 w (v x x) + t
__device__ __forceinline__ F3 qt_apply(float w, F3 v, F3 t, F3 x) { return x * (w * w - dot3(v, v)) + v * (2.f * dot3(v, x)) + cross3(v, x) * (2.f * w) + t; }

__global__ void __launch_bounds__(256) k_flow_cyc_fwd(const float* __restrict__ xyz_next, const float* __restrict__ q, const float* __restrict__ t,
                                                       const float* __restrict__ K, const float* __restrict__ hxy, const float* __restrict__ xyz_cyc,
                                                       const float* __restrict__ xyz_t, long S, int spf, int D, float thresh, float* __restrict__ flow,
                                                       float* __restrict__ cyc) {
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (long)gridDim.x * blockDim.x) {
    const long m = s / spf, r = s / D;
    const float* qm = q + m * 4;
    const float* Km = K + m * 9;
    const F3 xc = qt_apply(qm[0], ld3(qm + 1), ld3(t + m * 3), ld3(xyz_next + s * 3));
    const float hx = Km[0] * xc.x + Km[1] * xc.y + Km[2] * xc.z, hy = Km[3] * xc.x + Km[4] * xc.y + Km[5] * xc.z, hz = Km[6] * xc.x + Km[7] * xc.y + Km[8] * xc.z;
    const float iz = 1.f / (hz + 1e-6f);
    const float fu = hx * iz - hxy[r * 3], fv = hy * iz - hxy[r * 3 + 1];
    bool valid = xc.z > 1e-6f;
    if (thresh >= 0.f) valid = valid && (sqrtf(fu * fu + fv * fv) < thresh);
    flow[s * 3] = fu; flow[s * 3 + 1] = fv; flow[s * 3 + 2] = valid ? 1.f : 0.f;
    if (cyc) {
      const F3 d = ld3(xyz_cyc + s * 3) - ld3(xyz_t + s * 3);
      cyc[s] = sqrtf(dot3(d, d));
    }
  }
}

// per-frame accumulators: [g_q (4) | g_t (3) | g_K (9)] = 16 floats per frame
__global__ void __launch_bounds__(256) k_flow_cyc_bwd(const float* __restrict__ xyz_next, const float* __restrict__ q, const float* __restrict__ t,
                                                       const float* __restrict__ K, const float* __restrict__ xyz_cyc, const float* __restrict__ xyz_t,
                                                       const float* __restrict__ g_flow, const float* __restrict__ g_cyc, long S, int spf,
                                                       float* __restrict__ g_xyz_next, float* __restrict__ g_pf, float* __restrict__ g_xyz_cyc,
                                                       float* __restrict__ g_xyz_t) {
  // a block works on a contiguous run of `chunk` samples of ONE frame (blockIdx.y = frame), so the 16 per-frame sums are reduced in
  // the block and leave as 16 atomics
  const long m = blockIdx.y;
  const long per = ((long)spf + gridDim.x - 1) / gridDim.x;
  const long s0 = m * spf + (long)blockIdx.x * per;
  long s1 = s0 + per;
  if (s1 > (m + 1) * (long)spf) s1 = (m + 1) * (long)spf;
  if (s1 > S) s1 = S;
  const float* qm = q + m * 4;
  const float* Km = K + m * 9;
  const float w = qm[0];
  const F3 v = ld3(qm + 1), tt = ld3(t + m * 3);
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (long s = s0 + threadIdx.x; s < s1; s += blockDim.x) {
    const F3 x = ld3(xyz_next + s * 3);
    const F3 xc = qt_apply(w, v, tt, x);
    const float hx = Km[0] * xc.x + Km[1] * xc.y + Km[2] * xc.z, hy = Km[3] * xc.x + Km[4] * xc.y + Km[5] * xc.z, hz = Km[6] * xc.x + Km[7] * xc.y + Km[8] * xc.z;
    const float iz = 1.f / (hz + 1e-6f);
    const float gu = g_flow[s * 3], gv = g_flow[s * 3 + 1];
    // u = hx * iz, v = hy * iz
    const F3 gh = {gu * iz, gv * iz, -(gu * hx + gv * hy) * iz * iz};
    const F3 gxc = {Km[0] * gh.x + Km[3] * gh.y + Km[6] * gh.z, Km[1] * gh.x + Km[4] * gh.y + Km[7] * gh.z, Km[2] * gh.x + Km[5] * gh.y + Km[8] * gh.z};
    // rotation adjoint: R^T g = (w^2 - v.v) g + 2 (v.g) v - 2 w (v x g)
    st3(g_xyz_next + s * 3, gxc * (w * w - dot3(v, v)) + v * (2.f * dot3(v, gxc)) - cross3(v, gxc) * (2.f * w));
    acc[0] += 2.f * w * dot3(x, gxc) + 2.f * dot3(cross3(v, x), gxc);
    const F3 gvq = v * (-2.f * dot3(x, gxc)) + x * (2.f * dot3(v, gxc)) + gxc * (2.f * dot3(v, x)) + cross3(x, gxc) * (2.f * w);
    acc[1] += gvq.x; acc[2] += gvq.y; acc[3] += gvq.z;
    acc[4] += gxc.x; acc[5] += gxc.y; acc[6] += gxc.z;
    acc[7] += gh.x * xc.x; acc[8] += gh.x * xc.y; acc[9] += gh.x * xc.z;
    acc[10] += gh.y * xc.x; acc[11] += gh.y * xc.y; acc[12] += gh.y * xc.z;
    acc[13] += gh.z * xc.x; acc[14] += gh.z * xc.y; acc[15] += gh.z * xc.z;
    if (g_xyz_cyc) {
      const F3 d = ld3(xyz_cyc + s * 3) - ld3(xyz_t + s * 3);
      const float n = sqrtf(dot3(d, d));
      const F3 gd = n > 0.f ? d * (g_cyc[s] / n) : F3{0.f, 0.f, 0.f};  // zero subgradient at a zero distance, like torch's norm
      st3(g_xyz_cyc + s * 3, gd);
      st3(g_xyz_t + s * 3, gd * -1.f);
    }
  }
  __shared__ float red[4][16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float a = wave_sum(acc[i]);
    if (lane == 0) red[wid][i] = a;
  }
  __syncthreads();
  if (threadIdx.x < 16) atomicAdd(g_pf + m * 16 + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// VolSDF density (nerf.py:186-192): (0.5 + 0.5 sign(s) expm1(-|s| ib)) ib
__global__ void __launch_bounds__(256) k_volsdf_fwd(const float* __restrict__ sdf, const float* __restrict__ ibeta, long S, float* __restrict__ out) {
  const float ib = ibeta[0];
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (long)gridDim.x * blockDim.x) {
    const float x = sdf[s];
    const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
    out[s] = (0.5f + 0.5f * sg * expm1f(-fabsf(x) * ib)) * ib;
  }
}
__global__ void __launch_bounds__(256) k_volsdf_bwd(const float* __restrict__ sdf, const float* __restrict__ ibeta, const float* __restrict__ g, long S,
                                                     float* __restrict__ g_sdf, float* __restrict__ g_ibeta) {
  const float ib = ibeta[0];
  float acc = 0.f;
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (long)gridDim.x * blockDim.x) {
    const float x = sdf[s], a = fabsf(x);
    const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
    const float e = expf(-a * ib);                      // expm1(z) + 1
    // out = (0.5 + 0.5 sg (e - 1)) ib ;  d out / d x = 0.5 sg e (-sg ib) ib = -0.5 sg^2 e ib^2 ;  d out / d ib = 0.5 + 0.5 sg (e - 1) - 0.5 sg e a ib
    g_sdf[s] = g[s] * (-0.5f * sg * sg * e * ib * ib);
    acc += g[s] * (0.5f + 0.5f * sg * (e - 1.f) - 0.5f * sg * e * a * ib);
  }
  __shared__ float red[4];
  const float a = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0 && g_ibeta) atomicAdd(g_ibeta, red[0] + red[1] + red[2] + red[3]);
}

}  // namespace lab4d
using namespace lab4d;

static int grid1(long n) { long g = (n + 255) / 256; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

extern "C" int lab4d_flow_cyc_forward(const float* xyz_next, const float* q, const float* t, const float* K, const float* hxy, const float* xyz_cyc,
                                      const float* xyz_t, long S, int spf, int D, float flow_thresh, float* flow, float* cyc, void* stream) {
  LAB4D_REQUIRE(xyz_next && q && t && K && hxy && flow, "flow_cyc_forward: null pointer");
  LAB4D_REQUIRE((xyz_cyc == nullptr) == (cyc == nullptr) && (xyz_cyc == nullptr) == (xyz_t == nullptr), "flow_cyc_forward: cycle inputs / output go together");
  LAB4D_REQUIRE(spf > 0 && D > 0 && spf % D == 0, "flow_cyc_forward: samples per frame must be a multiple of samples per ray");
  if (S == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_flow_cyc_fwd, dim3(grid1(S)), dim3(256), 0, (hipStream_t)stream, xyz_next, q, t, K, hxy, xyz_cyc, xyz_t, S, spf, D, flow_thresh, flow, cyc);
  return check_launch("flow_cyc_forward");
}

extern "C" int lab4d_flow_cyc_backward(const float* xyz_next, const float* q, const float* t, const float* K, const float* xyz_cyc, const float* xyz_t,
                                       const float* g_flow, const float* g_cyc, long S, int spf, int M, float* g_xyz_next, float* g_per_frame,
                                       float* g_xyz_cyc, float* g_xyz_t, void* stream) {
  LAB4D_REQUIRE(xyz_next && q && t && K && g_flow && g_xyz_next && g_per_frame, "flow_cyc_backward: null pointer");
  LAB4D_REQUIRE((g_xyz_cyc == nullptr) == (g_xyz_t == nullptr) && (g_xyz_cyc == nullptr || (xyz_cyc && xyz_t && g_cyc)), "flow_cyc_backward: cycle arguments go together");
  LAB4D_REQUIRE(spf > 0 && M > 0 && (long)M * spf >= S, "flow_cyc_backward: M * spf < S");
  hipStream_t st = (hipStream_t)stream;
  // before the early return: an empty chunk still hands g_per_frame back as the (zero) gradients of q / t / K
  if (int e = zero_async(g_per_frame, (size_t)M * 16 * sizeof(float), st)) return e;
  if (S == 0) return LAB4D_OK;
  int bx = (int)((spf + 16383) / 16384);
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(k_flow_cyc_bwd, dim3(bx, M), dim3(256), 0, st, xyz_next, q, t, K, xyz_cyc, xyz_t, g_flow, g_cyc, S, spf, g_xyz_next, g_per_frame, g_xyz_cyc,
                     g_xyz_t);
  return check_launch("flow_cyc_backward");
}

extern "C" int lab4d_volsdf_forward(const float* sdf, const float* ibeta, long S, float* density, void* stream) {
  LAB4D_REQUIRE(sdf && ibeta && density, "volsdf_forward: null pointer");
  if (S == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_volsdf_fwd, dim3(grid1(S)), dim3(256), 0, (hipStream_t)stream, sdf, ibeta, S, density);
  return check_launch("volsdf_forward");
}

extern "C" int lab4d_volsdf_backward(const float* sdf, const float* ibeta, const float* g_density, long S, float* g_sdf, float* g_ibeta, void* stream) {
  LAB4D_REQUIRE(sdf && ibeta && g_density && g_sdf, "volsdf_backward: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (g_ibeta)
    if (int e = zero_async(g_ibeta, sizeof(float), st)) return e;
  if (S == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_volsdf_bwd, dim3(grid1(S) > 1024 ? 1024 : grid1(S)), dim3(256), 0, st, sdf, ibeta, g_density, S, g_sdf, g_ibeta);
  return check_launch("volsdf_backward");
}
