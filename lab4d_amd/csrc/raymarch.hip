// Ray sampling kernels: sample_cam_rays (+ cam_to_field), sample_pdf, depth merge.
// Reference: lab4d/utils/render_utils.py:8-56,187-233; lab4d/nnutils/nerf.py:686-738,821-844.
// All HBM-bound: one thread per sample (coalesced 12-byte vector stores) for the forward, one wave
// per ray with lanes over samples for the adjoint (coalesced reads, wave reductions, per-frame
// accumulation through LDS then a handful of atomics per block).
#include "common.hpp"
#include "sample_pdf_math.hpp"

namespace lab4d {

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* p, F3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ F3 add3(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 mul3(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }

// vector part of q (0,p) conj(q) for a (not necessarily unit) quaternion q = (w, v):
//   (w^2 - v.v) p + 2 (v.p) v + 2 w (v x p)       [quat_transform.py:255-272]
__device__ __forceinline__ F3 qrot(float w, F3 v, F3 p) {
  const float a = w * w - dot3(v, v), b = 2.f * dot3(v, p), c = 2.f * w;
  const F3 x = cross3(v, p);
  return {a * p.x + b * v.x + c * x.x, a * p.y + b * v.y + c * x.y, a * p.z + b * v.z + c * x.z};
}
// transpose of the linear map p -> qrot(q,p)
__device__ __forceinline__ F3 qrot_t(float w, F3 v, F3 g) {
  const float a = w * w - dot3(v, v), b = 2.f * dot3(v, g), c = -2.f * w;
  const F3 x = cross3(v, g);
  return {a * g.x + b * v.x + c * x.x, a * g.y + b * v.y + c * x.y, a * g.z + b * v.z + c * x.z};
}

__device__ __forceinline__ float depth_at(const float* near_far, const float* depth_in, int m, long ray, int d, int D) {
  if (depth_in) return depth_in[ray * D + d];
  // near*(1-z) + far*z with z = linspace(0,1,D)[d]  (render_utils.py:29-31; torch.linspace is
  // evaluated from both ends: start + i*step below the midpoint, end - (D-1-i)*step above)
  const float step = 1.0f / (float)(D - 1);
  const float z = (d < D / 2) ? __fmul_rn(step, (float)d) : __fsub_rn(1.0f, __fmul_rn(step, (float)(D - 1 - d)));
  return __fadd_rn(__fmul_rn(near_far[2 * m], __fsub_rn(1.0f, z)), __fmul_rn(near_far[2 * m + 1], z));
}

__global__ void __launch_bounds__(256) k_ray_samples_fwd(const float* __restrict__ hxy, const float* __restrict__ Kinv,
                                                          const float* __restrict__ near_far, const float* __restrict__ depth_in,
                                                          const float* __restrict__ cq, const float* __restrict__ ct, int M, int N, int D,
                                                          float* __restrict__ xyz_cam, float* __restrict__ dir_cam,
                                                          float* __restrict__ deltas, float* __restrict__ depth,
                                                          float* __restrict__ xyz_field, float* __restrict__ dir_field) {
  const long S = (long)M * N * D;
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (long)gridDim.x * blockDim.x) {
    const long ray = s / D;
    const int d = (int)(s - ray * D);
    const int m = (int)(ray / N);
    const F3 h = ld3(hxy + ray * 3);
    const float* K = Kinv + m * 9;
    const F3 dir = {K[0] * h.x + K[1] * h.y + K[2] * h.z, K[3] * h.x + K[4] * h.y + K[5] * h.z, K[6] * h.x + K[7] * h.y + K[8] * h.z};
    const float nrm = sqrtf(dot3(dir, dir));
    const float z = depth_at(near_far, depth_in, m, ray, d, D);
    // interval to the next sample, the last one repeats the previous interval (render_utils.py:47-50)
    float dz;
    if (d + 1 < D) dz = depth_at(near_far, depth_in, m, ray, d + 1, D) - z;
    else dz = z - depth_at(near_far, depth_in, m, ray, d - 1, D);
    const F3 p = mul3(dir, z);
    const F3 u = mul3(dir, 1.0f / nrm);
    if (xyz_cam) st3(xyz_cam + s * 3, p);
    if (dir_cam) st3(dir_cam + s * 3, u);
    if (deltas) deltas[s] = dz * nrm;
    if (depth) depth[s] = z;
    if (cq) {
      const float w = cq[4 * m];
      const F3 v = ld3(cq + 4 * m + 1);
      if (xyz_field) st3(xyz_field + s * 3, add3(qrot(w, v, p), ld3(ct + 3 * m)));
      if (dir_field) st3(dir_field + s * 3, qrot(w, v, u));
    }
  }
}

// One wave per ray, lanes over samples; a block (4 waves) walks RAYS_PER_WAVE rays per wave of one
// frame and accumulates the 16 per-frame gradients in registers -> LDS -> atomics.
constexpr int kRaysPerWave = 16;
__global__ void __launch_bounds__(256) k_ray_samples_bwd(const float* __restrict__ hxy, const float* __restrict__ Kinv,
                                                          const float* __restrict__ near_far, const float* __restrict__ depth_in,
                                                          const float* __restrict__ cq, const float* __restrict__ ct, int M, int N, int D,
                                                          const float* __restrict__ g_xyz_cam, const float* __restrict__ g_dir_cam,
                                                          const float* __restrict__ g_deltas, const float* __restrict__ g_xyz_field,
                                                          const float* __restrict__ g_dir_field, float* __restrict__ g_Kinv,
                                                          float* __restrict__ g_q, float* __restrict__ g_t) {
  __shared__ float red[4][16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int blocks_per_frame = div_up_dev(N, 4 * kRaysPerWave);
  const int m = blockIdx.x / blocks_per_frame;
  const int n0 = (blockIdx.x - m * blocks_per_frame) * 4 * kRaysPerWave + wid * kRaysPerWave;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const float* K = Kinv + m * 9;
  float qw = 1.f; F3 qv = {0, 0, 0};
  if (cq) { qw = cq[4 * m]; qv = ld3(cq + 4 * m + 1); }
  for (int r = 0; r < kRaysPerWave; ++r) {
    const int n = n0 + r;
    if (n >= N) break;
    const long ray = (long)m * N + n;
    const F3 h = ld3(hxy + ray * 3);
    const F3 dir = {K[0] * h.x + K[1] * h.y + K[2] * h.z, K[3] * h.x + K[4] * h.y + K[5] * h.z, K[6] * h.x + K[7] * h.y + K[8] * h.z};
    const float nrm = sqrtf(dot3(dir, dir));
    F3 gdir = {0, 0, 0};   // grad wrt un-normalised dir from xyz = dir*z
    F3 gu = {0, 0, 0};     // grad wrt normalised direction
    float gnrm = 0.f;      // grad wrt |dir| from deltas
    F3 gt = {0, 0, 0};
    float gqw = 0.f; F3 gqv = {0, 0, 0};
    for (int d = lane; d < D; d += 64) {
      const long s = ray * D + d;
      const float z = depth_at(near_far, depth_in, m, ray, d, D);
      float dz;
      if (d + 1 < D) dz = depth_at(near_far, depth_in, m, ray, d + 1, D) - z;
      else dz = z - depth_at(near_far, depth_in, m, ray, d - 1, D);
      F3 gp = {0, 0, 0};
      if (g_xyz_cam) gp = ld3(g_xyz_cam + s * 3);
      if (cq && g_xyz_field) {
        const F3 g = ld3(g_xyz_field + s * 3);
        const F3 p = mul3(dir, z);
        gp = add3(gp, qrot_t(qw, qv, g));
        gt = add3(gt, g);
        // d qrot / d q contracted with g
        const float gp_ = dot3(g, p), gv_ = dot3(g, qv), vp_ = dot3(qv, p);
        const F3 pxg = cross3(p, g);
        gqw += 2.f * qw * gp_ + 2.f * dot3(g, cross3(qv, p));
        gqv = add3(gqv, {-2.f * qv.x * gp_ + 2.f * p.x * gv_ + 2.f * vp_ * g.x + 2.f * qw * pxg.x,
                         -2.f * qv.y * gp_ + 2.f * p.y * gv_ + 2.f * vp_ * g.y + 2.f * qw * pxg.y,
                         -2.f * qv.z * gp_ + 2.f * p.z * gv_ + 2.f * vp_ * g.z + 2.f * qw * pxg.z});
      }
      gdir = add3(gdir, mul3(gp, z));
      if (g_deltas) gnrm += g_deltas[s] * dz;
      if (g_dir_cam) gu = add3(gu, ld3(g_dir_cam + s * 3));
      if (cq && g_dir_field) {
        const F3 g = ld3(g_dir_field + s * 3);
        const F3 u = mul3(dir, 1.0f / nrm);
        gu = add3(gu, qrot_t(qw, qv, g));
        const float gp_ = dot3(g, u), gv_ = dot3(g, qv), vp_ = dot3(qv, u);
        const F3 pxg = cross3(u, g);
        gqw += 2.f * qw * gp_ + 2.f * dot3(g, cross3(qv, u));
        gqv = add3(gqv, {-2.f * qv.x * gp_ + 2.f * u.x * gv_ + 2.f * vp_ * g.x + 2.f * qw * pxg.x,
                         -2.f * qv.y * gp_ + 2.f * u.y * gv_ + 2.f * vp_ * g.y + 2.f * qw * pxg.y,
                         -2.f * qv.z * gp_ + 2.f * u.z * gv_ + 2.f * vp_ * g.z + 2.f * qw * pxg.z});
      }
    }
    // u = dir/nrm: gdir += (gu - u (u.gu))/nrm ; nrm = |dir|: gdir += gnrm * u
    const F3 u = mul3(dir, 1.0f / nrm);
    const float ugu = dot3(u, gu);
    gdir = add3(gdir, add3(mul3(add3(gu, mul3(u, -ugu)), 1.0f / nrm), mul3(u, gnrm)));
    acc[0] += gdir.x * h.x; acc[1] += gdir.x * h.y; acc[2] += gdir.x * h.z;
    acc[3] += gdir.y * h.x; acc[4] += gdir.y * h.y; acc[5] += gdir.y * h.z;
    acc[6] += gdir.z * h.x; acc[7] += gdir.z * h.y; acc[8] += gdir.z * h.z;
    acc[9] += gqw; acc[10] += gqv.x; acc[11] += gqv.y; acc[12] += gqv.z;
    acc[13] += gt.x; acc[14] += gt.y; acc[15] += gt.z;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float v = wave_sum(acc[i]);
    if (lane == 0) red[wid][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int i = threadIdx.x;
    const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (i < 9) { if (g_Kinv) atomicAdd(g_Kinv + m * 9 + i, v); }
    else if (i < 13) { if (g_q) atomicAdd(g_q + m * 4 + (i - 9), v); }
    else { if (g_t) atomicAdd(g_t + m * 3 + (i - 13), v); }
  }
}

// sample_pdf: one thread per ray (csrc/sample_pdf_math.hpp holds the per-ray arithmetic).
// u_in: caller-drawn uniforms (R, n_imp), sorted ascending per ray (det=False: the host draws torch.rand, sorts, and un-sorts
// the result -- searchsorted is a per-element operation, so the order of the queries is immaterial), or NULL for linspace.
__global__ void __launch_bounds__(256) k_sample_pdf(const float* __restrict__ bins, const float* __restrict__ weights, int R,
                                                     int n_w, int n_imp, float eps, const float* __restrict__ u_in,
                                                     float* __restrict__ samples, int64_t* __restrict__ inds) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  lab4d_pdf::sample_pdf_ray(bins + (long)r * (n_w + 1), weights + (long)r * n_w, n_w, n_imp, eps, u_in ? u_in + (long)r * n_imp : nullptr,
                            samples + (long)r * n_imp, inds + (long)r * n_imp);
}

// sorted(cat(a, b)) per ray: merge two (nearly) sorted runs, then one insertion pass to repair the
// rare rounding inversions so that the result equals torch.sort on the concatenation (nerf.py:731).
__global__ void __launch_bounds__(256) k_sort_depth(const float* __restrict__ a, int na, const float* __restrict__ b, int nb, int R,
                                                     float* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* pa = a + (long)r * na;
  const float* pb = b + (long)r * nb;
  float* o = out + (long)r * (na + nb);
  int i = 0, j = 0, k = 0;
  while (i < na && j < nb) o[k++] = (pb[j] < pa[i]) ? pb[j++] : pa[i++];
  while (i < na) o[k++] = pa[i++];
  while (j < nb) o[k++] = pb[j++];
  for (int x = 1; x < na + nb; ++x) {
    const float v = o[x];
    int y = x - 1;
    while (y >= 0 && o[y] > v) { o[y + 1] = o[y]; --y; }
    o[y + 1] = v;
  }
}

}  // namespace lab4d
using namespace lab4d;

extern "C" int lab4d_ray_samples_forward(const float* hxy, const float* Kinv, const float* near_far, const float* depth_in,
                                         const float* cq, const float* ct, int M, int N, int D, float* xyz_cam,
                                         float* dir_cam, float* deltas, float* depth, float* xyz_field, float* dir_field,
                                         void* stream) {
  LAB4D_REQUIRE(hxy && Kinv && (near_far || depth_in), "ray_samples_forward: null input");
  LAB4D_REQUIRE(M >= 0 && N >= 0 && D >= 2, "ray_samples_forward: need D >= 2 (got M=%d N=%d D=%d)", M, N, D);
  LAB4D_REQUIRE((cq == nullptr) == (ct == nullptr), "ray_samples_forward: cam2field q and t must be given together");
  const long S = (long)M * N * D;
  if (S == 0) return LAB4D_OK;
  int grid = div_up(S, 256); if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_ray_samples_fwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, hxy, Kinv, near_far, depth_in, cq, ct,
                     M, N, D, xyz_cam, dir_cam, deltas, depth, xyz_field, dir_field);
  return check_launch("ray_samples_forward");
}

extern "C" int lab4d_ray_samples_backward(const float* hxy, const float* Kinv, const float* near_far, const float* depth_in,
                                          const float* cq, const float* ct, int M, int N, int D, const float* g_xyz_cam,
                                          const float* g_dir_cam, const float* g_deltas, const float* g_xyz_field,
                                          const float* g_dir_field, float* g_Kinv, float* g_q, float* g_t, void* stream) {
  LAB4D_REQUIRE(hxy && Kinv && (near_far || depth_in), "ray_samples_backward: null input");
  LAB4D_REQUIRE(M >= 0 && N >= 0 && D >= 2, "ray_samples_backward: need D >= 2");
  if ((long)M * N * D == 0) return LAB4D_OK;
  const int blocks_per_frame = div_up(N, 4 * kRaysPerWave);
  hipLaunchKernelGGL(k_ray_samples_bwd, dim3(M * blocks_per_frame), dim3(256), 0, (hipStream_t)stream, hxy, Kinv, near_far,
                     depth_in, cq, ct, M, N, D, g_xyz_cam, g_dir_cam, g_deltas, g_xyz_field, g_dir_field, g_Kinv, g_q, g_t);
  return check_launch("ray_samples_backward");
}

extern "C" int lab4d_sample_pdf(const float* bins, const float* weights, int R, int n_w, int n_imp, float eps, float* samples,
                                int64_t* inds, void* stream) {
  return lab4d_sample_pdf_u(bins, weights, nullptr, R, n_w, n_imp, eps, samples, inds, stream);
}

extern "C" int lab4d_sample_pdf_u(const float* bins, const float* weights, const float* u_sorted, int R, int n_w, int n_imp, float eps,
                                  float* samples, int64_t* inds, void* stream) {
  LAB4D_REQUIRE(bins && weights && samples && inds, "sample_pdf: null pointer");
  LAB4D_REQUIRE(n_w >= 1 && n_imp >= 2, "sample_pdf: need n_w >= 1 and n_imp >= 2");
  if (R == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_sample_pdf, dim3(div_up(R, 256)), dim3(256), 0, (hipStream_t)stream, bins, weights, R, n_w, n_imp, eps,
                     u_sorted, samples, inds);
  return check_launch("sample_pdf");
}

extern "C" int lab4d_sort_depth(const float* a, int na, const float* b, int nb, int R, float* out, void* stream) {
  LAB4D_REQUIRE(a && b && out, "sort_depth: null pointer");
  if (R == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_sort_depth, dim3(div_up(R, 256)), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, R, out);
  return check_launch("sort_depth");
}
