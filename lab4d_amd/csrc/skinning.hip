// Linear-blend skinning with dual quaternions, forward + adjoint (fp32 VALU kernels).
// Contract and reference citations: include/lab4d_skin.h.
#include "common.hpp"

namespace lab4d {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 ldv3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void stv3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// vector part of q (0,p) conj(q), q = (w, v) not necessarily unit (quat_transform.py:255-272)
__device__ __forceinline__ V3 qrot(float w, V3 v, V3 p) { return p * (w * w - dot(v, v)) + v * (2.f * dot(v, p)) + cross(v, p) * (2.f * w); }
__device__ __forceinline__ V3 qrot_t(float w, V3 v, V3 g) { return g * (w * w - dot(v, v)) + v * (2.f * dot(v, g)) - cross(v, g) * (2.f * w); }
// gradient of g . qrot(q, p) wrt (w, v)
__device__ __forceinline__ void qrot_gq(float w, V3 v, V3 p, V3 g, float& gw, V3& gv) {
  const float gp = dot(g, p), gvv = dot(g, v), vp = dot(v, p);
  gw = 2.f * w * gp + 2.f * dot(g, cross(v, p));
  gv = v * (-2.f * gp) + p * (2.f * gvv) + g * (2.f * vp) + cross(p, g) * (2.f * w);
}


// ---------------------------------------------------------------------------------------------
// (S, C) fp32 rows staged through LDS.  One thread per sample reading its own C-float row straight from global memory touches one
// 128-byte line per lane and instruction (a wave's 64 rows span 64*C*4 bytes; every line is re-fetched for each of the C column
// steps once 16 waves per CU have pushed it out of the 32 KiB L1; partial-line writes are worse).  Instead a block's 256
// consecutive rows -- ONE contiguous 256*C*4-byte region -- move between global memory and LDS with 16-byte-per-lane accesses,
// and a thread reads / writes its row in LDS (row stride C | 1 floats: odd, hence bank-conflict-free).
// ---------------------------------------------------------------------------------------------
template <int C>
struct RowTile {
  static constexpr int CP = C | 1;            // odd row stride (floats)
  static constexpr int FLOATS = 256 * CP;
  // rows [s0, s0 + n) of g (row-major, C floats per row) -> tile; block-cooperative, all 256 threads call it.
  // Full tiles issue ALL their global loads before the first LDS write (a load -> write loop pays one memory latency per pass:
  // 19 passes for the 75-float rows, at two blocks per CU that was the whole kernel time).
  static __device__ __forceinline__ void load(const float* __restrict__ g, long s0, int n, float* __restrict__ tile) {
    const float* src = g + s0 * C;
    const int total = n * C;
    if ((((size_t)src) & 15) == 0) {
      constexpr int TRIPS = (256 * C + 1023) / 1024;
      if (n == 256) {
        float4 v[TRIPS];
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
          const int e = 4 * (int)threadIdx.x + 1024 * t;
          if (e + 3 < 256 * C) v[t] = *reinterpret_cast<const float4*>(src + e);
        }
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
          const int e = 4 * (int)threadIdx.x + 1024 * t;
          if (e + 3 < 256 * C) {
            const float vv[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int r = (e + k) / C; tile[r * CP + (e + k) - r * C] = vv[k]; }
          }
        }
        static_assert((256 * C) % 4 == 0, "a full tile is a whole number of float4");
        return;
      }
      for (int e = 4 * (int)threadIdx.x; e < total; e += 1024) {
        if (e + 3 < total) {
          const float4 v = *reinterpret_cast<const float4*>(src + e);
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) { const int r = (e + k) / C; tile[r * CP + (e + k) - r * C] = vv[k]; }
        } else {
          for (int k = 0; e + k < total; ++k) { const int r = (e + k) / C; tile[r * CP + (e + k) - r * C] = src[e + k]; }
        }
      }
    } else {
      for (int e = threadIdx.x; e < total; e += 256) { const int r = e / C; tile[r * CP + e - r * C] = src[e]; }
    }
  }
  // add != 0: the rows are added to what g holds (read and written in the same coalesced float4 pieces)
  static __device__ __forceinline__ void store(float* __restrict__ g, long s0, int n, const float* __restrict__ tile, int add = 0) {
    float* dst = g + s0 * C;
    const int total = n * C;
    if ((((size_t)dst) & 15) == 0) {
      for (int e = 4 * (int)threadIdx.x; e < total; e += 1024) {
        if (e + 3 < total) {
          float vv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) { const int r = (e + k) / C; vv[k] = tile[r * CP + (e + k) - r * C]; }
          if (add) {
            const float4 o = *reinterpret_cast<const float4*>(dst + e);
            vv[0] += o.x; vv[1] += o.y; vv[2] += o.z; vv[3] += o.w;
          }
          *reinterpret_cast<float4*>(dst + e) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
          for (int k = 0; e + k < total; ++k) { const int r = (e + k) / C; dst[e + k] = tile[r * CP + (e + k) - r * C] + (add ? dst[e + k] : 0.f); }
        }
      }
    } else {
      for (int e = threadIdx.x; e < total; e += 256) { const int r = e / C; dst[e] = tile[r * CP + e - r * C] + (add ? dst[e] : 0.f); }
    }
  }
};
// frame of sample s: 32-bit division (a 64-bit division by a runtime value costs ~100 instructions per thread)
__device__ __forceinline__ int frame_of(long s, int spf) { return (int)((unsigned)s / (unsigned)spf); }

// ---------------------------------------------------------------------------------------------
// bone coordinates
// ---------------------------------------------------------------------------------------------
// inverse bone transform of bone-to-object dq (r = (w,v), d = (dw,dv)):  rotation q = conj(r),
// translation t = 2 vec(conj(d) r) = 2 (dw v - w dv + v x dv)      (transforms.py:19-24, quat_transform.py:337-344,441-465)
__device__ __forceinline__ V3 bone_apply(float w, V3 v, float dw, V3 dv, V3 x) {
  const V3 t = (v * dw - dv * w + cross(v, dv)) * 2.f;
  return qrot(w, v * -1.f, x) + t;
}

// One thread per sample; the (S,3B) rows leave through an LDS tile as coalesced float4 stores (RowTile).  UNI (spf % 256 == 0): the
// block's samples share one frame, so the bone tables are wave-uniform and come through SCALAR loads -- the first version gave every
// (sample, bone) its own thread and fetched the 11 table floats of its bone with per-lane gather loads: 14 vector-memory
// instructions per 12 bytes of output, bound by the texture addresser (2.4 TB/s).
template <int B, bool UNI>
__global__ void __launch_bounds__(256) k_bone_fwd(const float* __restrict__ xyz, const float* __restrict__ ar, const float* __restrict__ ad,
                                                   const float* __restrict__ gauss, long S, int spf, float* __restrict__ out) {
  using T = RowTile<3 * B>;
  __shared__ float tile[T::FLOATS];
  for (long s0 = (long)blockIdx.x * 256; s0 < S; s0 += (long)gridDim.x * 256) {
    const long rem = S - s0;
    const int n = (int)(rem < 256 ? rem : 256);
    if ((int)threadIdx.x < n) {
      const long s = s0 + threadIdx.x;
      const int m = UNI ? __builtin_amdgcn_readfirstlane(frame_of(s0, spf)) : frame_of(s, spf);
      const V3 x = ldv3(xyz + s * 3);
      float* row = tile + threadIdx.x * T::CP;
#pragma unroll 5
      for (int b = 0; b < B; ++b) {
        const float* r = ar + ((size_t)m * B + b) * 4;
        const float* d = ad + ((size_t)m * B + b) * 4;
        const V3 y = bone_apply(r[0], ldv3(r + 1), d[0], ldv3(d + 1), x);
        const V3 gs = ldv3(gauss + 3 * b);
        stv3(row + 3 * b, {y.x / gs.x, y.y / gs.y, y.z / gs.z});
      }
    }
    __syncthreads();
    T::store(out, s0, n, tile);
    __syncthreads();
  }
}

// g_xyz[s] = sum_b R_b^T (g_bone[s,b] / gauss_b): one thread per sample.
// FUSE (needs UNI): the per-frame Gram matrix of the parameter path, G[m] += g_bone^T [x,1] (3B x 4), is formed from the LDS tile the
// rows already sit in (thread (slice, column) walks every SLICES-th row; slices meet in LDS; block sums stay in LDS across tiles and
// leave as one atomic per output when the frame changes and at the end) -- before round 3 a torch.cat built [x,1] in HBM and
// k_gram_pf_rb read the (S,3B) gradient a second time.
template <int B, bool UNI, bool FUSE>
__global__ void __launch_bounds__(256) k_bone_bwd_x(const float* __restrict__ ar, const float* __restrict__ gauss,
                                                     const float* __restrict__ g_bone, long S, int spf, float* __restrict__ g_xyz,
                                                     const float* __restrict__ xyz, float* __restrict__ G) {
  static_assert(!FUSE || UNI, "the fused Gram reduction needs frame-uniform tiles");
  using T = RowTile<3 * B>;
  constexpr int C = 3 * B, SLICES = 256 / C;
  __shared__ float tile[T::FLOATS];
  __shared__ float aux[FUSE ? 256 * 3 : 4];  // the points of the tile ([x,1]: the 1 is implicit -- a fourth float per row would push the block
  __shared__ float lacc[FUSE ? C * 4 : 1];   // past 80 KiB of LDS, i.e. from two blocks per CU to one)
  int cur_m = -1;
  if (FUSE)
    for (int e = threadIdx.x; e < C * 4; e += 256) lacc[e] = 0.f;
  auto flush = [&](int m) {
    for (int e = threadIdx.x; e < C * 4; e += 256) {
      const float v = lacc[e];
      if (v != 0.f) atomicAdd(G + (size_t)m * C * 4 + e, v);
      lacc[e] = 0.f;
    }
  };
  for (long s0 = (long)blockIdx.x * 256; s0 < S; s0 += (long)gridDim.x * 256) {
    const long rem = S - s0;
    const int n = (int)(rem < 256 ? rem : 256);
    if (FUSE) {
      const int m_tile = __builtin_amdgcn_readfirstlane(frame_of(s0, spf));
      if (m_tile != cur_m) {
        if (cur_m >= 0) { __syncthreads(); flush(cur_m); }
        cur_m = m_tile;
      }
    }
    T::load(g_bone, s0, n, tile);
    if (FUSE && (int)threadIdx.x < n) {
      stv3(aux + threadIdx.x * 3, ldv3(xyz + (s0 + threadIdx.x) * 3));
    }
    __syncthreads();
    if ((int)threadIdx.x < n) {
      const long s = s0 + threadIdx.x;
      const int m = UNI ? __builtin_amdgcn_readfirstlane(frame_of(s0, spf)) : frame_of(s, spf);
      const float* row = tile + threadIdx.x * T::CP;
      V3 acc = {0, 0, 0};
#pragma unroll 5
      for (int b = 0; b < B; ++b) {
        const float* r = ar + ((size_t)m * B + b) * 4;
        const V3 gs = ldv3(gauss + 3 * b);
        const V3 g = ldv3(row + 3 * b);
        acc = acc + qrot_t(r[0], ldv3(r + 1) * -1.f, {g.x / gs.x, g.y / gs.y, g.z / gs.z});
      }
      stv3(g_xyz + s * 3, acc);
    }
    if (FUSE) {
      const int t = threadIdx.x, sl = t / C, i = t - sl * C;
      const bool act = sl < SLICES;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      if (act)
        for (int k = sl; k < n; k += SLICES) {
          const float a = tile[k * T::CP + i];
          const V3 b = ldv3(aux + k * 3);
          a0 += a * b.x; a1 += a * b.y; a2 += a * b.z; a3 += a;
        }
      __syncthreads();  // the tile's rows are consumed: it becomes the scratch of the slice reduction
      if (act) { float* o = tile + (sl * C + i) * 4; o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; }
      __syncthreads();
      for (int e = t; e < C * 4; e += 256) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < SLICES; ++q) v += tile[q * C * 4 + e];
        lacc[e] += v;
      }
    }
    __syncthreads();
  }
  if (FUSE && cur_m >= 0) flush(cur_m);
}

// per-frame / per-bone parameter gradients: block = (frame m, chunk of samples), loop bones outermost
// so that each (sample,bone) gradient element is read exactly once and the 11 reductions per bone are
// amortised over the block's samples.
template <int B>
__global__ void __launch_bounds__(256) k_bone_bwd_p(const float* __restrict__ xyz, const float* __restrict__ ar, const float* __restrict__ ad,
                                                     const float* __restrict__ gauss, const float* __restrict__ g_bone, long S, int spf,
                                                     int chunk, float* __restrict__ g_ar, float* __restrict__ g_ad, float* __restrict__ g_gauss) {
  __shared__ float red[4][11];
  const int m = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long f0 = (long)m * spf, f1 = min(S, f0 + spf);
  const long c0 = f0 + (long)blockIdx.x * chunk, c1 = min(f1, c0 + chunk);
  if (c0 >= c1) return;
  for (int b = 0; b < B; ++b) {
    const float* r = ar + ((size_t)m * B + b) * 4;
    const float* d = ad + ((size_t)m * B + b) * 4;
    const float w = r[0], dw = d[0];
    const V3 v = ldv3(r + 1), dv = ldv3(d + 1), gs = ldv3(gauss + 3 * b);
    float a[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) a[k] = 0.f;
    for (long s = c0 + threadIdx.x; s < c1; s += 256) {
      const V3 x = ldv3(xyz + s * 3);
      const V3 go = ldv3(g_bone + (s * B + b) * 3);
      const V3 y = bone_apply(w, v, dw, dv, x);
      const V3 gy = {go.x / gs.x, go.y / gs.y, go.z / gs.z};
      // out = y / gauss
      a[8] -= go.x * y.x / (gs.x * gs.x); a[9] -= go.y * y.y / (gs.y * gs.y); a[10] -= go.z * y.z / (gs.z * gs.z);
      // rotation part: q = (w, -v)
      float gqw; V3 gqv;
      qrot_gq(w, v * -1.f, x, gy, gqw, gqv);
      // translation part t = 2 (dw v - w dv + v x dv)
      const float g_w = gqw - 2.f * dot(dv, gy);
      const V3 g_v = gqv * -1.f + gy * (2.f * dw) + cross(dv, gy) * 2.f;
      const float g_dw = 2.f * dot(v, gy);
      const V3 g_dv = gy * (-2.f * w) + cross(gy, v) * 2.f;
      a[0] += g_w; a[1] += g_v.x; a[2] += g_v.y; a[3] += g_v.z;
      a[4] += g_dw; a[5] += g_dv.x; a[6] += g_dv.y; a[7] += g_dv.z;
    }
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float t = wave_sum(a[k]);
      if (lane == 0) red[wid][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 11) {
      const int k = threadIdx.x;
      const float t = red[0][k] + red[1][k] + red[2][k] + red[3][k];
      if (k < 4) atomicAdd(g_ar + ((size_t)m * B + b) * 4 + k, t);
      else if (k < 8) atomicAdd(g_ad + ((size_t)m * B + b) * 4 + (k - 4), t);
      else atomicAdd(g_gauss + 3 * b + (k - 8), t);
    }
    __syncthreads();
  }
}

// Parameter gradients of the bone-coordinate map from its per-frame Gram matrix.  out[s,b,:] = (R_b x_s + t_b) / gauss_b
// is affine in the point, so with G[m,b,k,j] = sum_{s in frame m} g[s,b,k] * [x_s, 1]_j (lab4d_gram_per_frame) every
// per-sample adjoint of k_bone_bwd_p -- each bilinear in (g, [x,1]) -- collapses to one evaluation per (frame, bone):
// the rotation adjoint is qrot_gq applied to the three unit points with the Gram columns as gradients, the translation
// adjoint uses the homogeneous column.  One thread per (m,b); g_gauss is summed over frames with atomics.
__global__ void __launch_bounds__(64) k_bone_param_from_gram(const float* __restrict__ ar, const float* __restrict__ ad, const float* __restrict__ gauss,
                                                              const float* __restrict__ G, int M, int B, float* __restrict__ g_ar,
                                                              float* __restrict__ g_ad, float* __restrict__ g_gauss) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * B) return;
  const int b = i % B;
  const float* r = ar + (size_t)i * 4;
  const float* d = ad + (size_t)i * 4;
  const float w = r[0], dw = d[0];
  const V3 v = ldv3(r + 1), dv = ldv3(d + 1), gs = ldv3(gauss + 3 * b);
  const float* g = G + (size_t)i * 12;  // [k][j], k = output coordinate, j = 0..2 point coordinate, 3 = homogeneous
  // gauss: out_k = y_k / gs_k, y = R x + t  ->  d/d gs_k = - sum_j A[k][j] G[k][j] / gs_k^2, with A = [R | t] rows
  const V3 e0 = {1.f, 0.f, 0.f}, e1 = {0.f, 1.f, 0.f}, e2 = {0.f, 0.f, 1.f};
  const V3 t = (v * dw - dv * w + cross(v, dv)) * 2.f;
  const V3 c0 = qrot(w, v * -1.f, e0), c1 = qrot(w, v * -1.f, e1), c2 = qrot(w, v * -1.f, e2);  // columns of R
  const float yx = c0.x * g[0] + c1.x * g[1] + c2.x * g[2] + t.x * g[3];
  const float yy = c0.y * g[4] + c1.y * g[5] + c2.y * g[6] + t.y * g[7];
  const float yz = c0.z * g[8] + c1.z * g[9] + c2.z * g[10] + t.z * g[11];
  if (g_gauss) {
    atomicAdd(g_gauss + 3 * b + 0, -yx / (gs.x * gs.x));
    atomicAdd(g_gauss + 3 * b + 1, -yy / (gs.y * gs.y));
    atomicAdd(g_gauss + 3 * b + 2, -yz / (gs.z * gs.z));
  }
  if (g_ar || g_ad) {
    // gradient columns scaled by 1/gauss: gy_j = (G[0][j]/gs.x, G[1][j]/gs.y, G[2][j]/gs.z)
    const V3 gy0 = {g[0] / gs.x, g[4] / gs.y, g[8] / gs.z}, gy1 = {g[1] / gs.x, g[5] / gs.y, g[9] / gs.z};
    const V3 gy2 = {g[2] / gs.x, g[6] / gs.y, g[10] / gs.z}, gyh = {g[3] / gs.x, g[7] / gs.y, g[11] / gs.z};
    float qw0, qw1, qw2;
    V3 qv0, qv1, qv2;
    qrot_gq(w, v * -1.f, e0, gy0, qw0, qv0);
    qrot_gq(w, v * -1.f, e1, gy1, qw1, qv1);
    qrot_gq(w, v * -1.f, e2, gy2, qw2, qv2);
    const float gqw = qw0 + qw1 + qw2;
    const V3 gqv = qv0 + qv1 + qv2;
    // translation t = 2 (dw v - w dv + v x dv): adjoint with sum_s gy_s = gyh
    const float g_w = gqw - 2.f * dot(dv, gyh);
    const V3 g_v = gqv * -1.f + gyh * (2.f * dw) + cross(dv, gyh) * 2.f;
    const float g_dw = 2.f * dot(v, gyh);
    const V3 g_dv = gyh * (-2.f * w) + cross(gyh, v) * 2.f;
    if (g_ar) { float* o = g_ar + (size_t)i * 4; o[0] = g_w; o[1] = g_v.x; o[2] = g_v.y; o[3] = g_v.z; }
    if (g_ad) { float* o = g_ad + (size_t)i * 4; o[0] = g_dw; o[1] = g_dv.x; o[2] = g_dv.y; o[3] = g_dv.z; }
  }
}

// ---------------------------------------------------------------------------------------------
// skin weights + dual-quaternion blend
// ---------------------------------------------------------------------------------------------
template <int B>
struct Blend {
  float p[B];       // softmax weights
  float sg[B];      // hemisphere signs (+-1)
  float dl[B];      // delta = relu(raw) * 0.1
  int anchor;
  float lse_minus_max;
  float rw[4], dw4[4];  // un-normalised blended real / dual parts
  float inv;
};

// Per-frame affine form of the gaussian-scaled bone coordinates: c_b = Abar_b [x,1], Abar_b = diag(1/gauss_b) [R_b | t_b]
// (3x4, rows = output coordinate).  The blend kernels recompute c_b from it (12 FMAs per bone, no quaternion algebra and
// no divisions per sample) instead of re-reading the (S,3B) fp32 tensor that feeds the delta-skin MLP.
__global__ void __launch_bounds__(64) k_bone_affine(const float* __restrict__ ar, const float* __restrict__ ad, const float* __restrict__ gauss,
                                                     int M, int B, float* __restrict__ aff) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * B) return;
  const int b = i % B;
  const float* r = ar + (size_t)i * 4;
  const float* d = ad + (size_t)i * 4;
  const float w = r[0], dw = d[0];
  const V3 v = ldv3(r + 1), dv = ldv3(d + 1), gs = ldv3(gauss + 3 * b);
  const V3 e0 = {1.f, 0.f, 0.f}, e1 = {0.f, 1.f, 0.f}, e2 = {0.f, 0.f, 1.f};
  const V3 t = (v * dw - dv * w + cross(v, dv)) * 2.f;
  const V3 c0 = qrot(w, v * -1.f, e0), c1 = qrot(w, v * -1.f, e1), c2 = qrot(w, v * -1.f, e2);  // columns of R
  float* o = aff + (size_t)i * 12;
  o[0] = c0.x / gs.x; o[1] = c1.x / gs.x; o[2] = c2.x / gs.x; o[3] = t.x / gs.x;
  o[4] = c0.y / gs.y; o[5] = c1.y / gs.y; o[6] = c2.y / gs.y; o[7] = t.y / gs.y;
  o[8] = c0.z / gs.z; o[9] = c1.z / gs.z; o[10] = c2.z / gs.z; o[11] = t.z / gs.z;
}
struct Aff { float4 r0, r1, r2; };
__device__ __forceinline__ Aff ld_aff(const float* __restrict__ a) {
  const float4* p = reinterpret_cast<const float4*>(a);
  return {p[0], p[1], p[2]};
}
__device__ __forceinline__ V3 bone_coord(const Aff& A, V3 x) {
  return {A.r0.x * x.x + A.r0.y * x.y + A.r0.z * x.z + A.r0.w, A.r1.x * x.x + A.r1.y * x.y + A.r1.z * x.z + A.r1.w,
          A.r2.x * x.x + A.r2.y * x.y + A.r2.z * x.z + A.r2.w};
}

template <int B>
__device__ __forceinline__ void blend_forward(V3 x, const float* __restrict__ affm, const float* __restrict__ raw,
                                              const float* __restrict__ sr, const float* __restrict__ sd, Blend<B>& o) {
  float skin[B];
  float mx = -INFINITY;
  int am = 0;
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const V3 c = bone_coord(ld_aff(affm + 12 * b), x);
    const float dlt = fmaxf(raw[b], 0.f) * 0.1f;   // skinning.py:119
    o.dl[b] = dlt;
    skin[b] = -(dot(c, c) + dlt);                  // skinning.py:120
    if (skin[b] > mx) { mx = skin[b]; am = b; }    // first maximum, like torch.argmax
  }
  float sum = 0.f;
#pragma unroll
  for (int b = 0; b < B; ++b) { o.p[b] = expf(skin[b] - mx); sum += o.p[b]; }
  const float isum = 1.f / sum;
  o.anchor = am;
  o.lse_minus_max = logf(sum);  // logsumexp - max  (loss_utils.py:21-42)
  const float a0 = sr[4 * am], a1 = sr[4 * am + 1], a2 = sr[4 * am + 2], a3 = sr[4 * am + 3];
#pragma unroll
  for (int k = 0; k < 4; ++k) { o.rw[k] = 0.f; o.dw4[k] = 0.f; }
#pragma unroll
  for (int b = 0; b < B; ++b) {
    o.p[b] *= isum;
    const float* r = sr + 4 * b;
    const float* d = sd + 4 * b;
    const float sgn = (a0 * r[0] + a1 * r[1] + a2 * r[2] + a3 * r[3]) > 0.f ? 1.f : -1.f;  // geom_utils.py:66-70
    o.sg[b] = sgn;
    const float c = o.p[b] * sgn;
#pragma unroll
    for (int k = 0; k < 4; ++k) { o.rw[k] += c * r[k]; o.dw4[k] += c * d[k]; }
  }
  o.inv = 1.f / sqrtf(o.rw[0] * o.rw[0] + o.rw[1] * o.rw[1] + o.rw[2] * o.rw[2] + o.rw[3] * o.rw[3]);
}

// apply the normalised blended dq (r = (w,v), d = (dw,dv)):  out = qrot(r, x) + 2 (-dw v + w dv + v x dv)
__device__ __forceinline__ V3 dq_apply(float w, V3 v, float dw, V3 dv, V3 x) {
  return qrot(w, v, x) + (v * -dw + dv * w + cross(v, dv)) * 2.f;
}

template <int B, bool UNI>
__global__ void __launch_bounds__(256) k_blend_fwd(const float* __restrict__ xyz, const float* __restrict__ aff, const float* __restrict__ raw,
                                                    const float* __restrict__ sr, const float* __restrict__ sd, long S, int spf,
                                                    float* __restrict__ out, float* __restrict__ ent, float* __restrict__ dskin) {
  using T = RowTile<B>;
  __shared__ float tile[T::FLOATS];
  for (long s0 = (long)blockIdx.x * 256; s0 < S; s0 += (long)gridDim.x * 256) {
    const long rem = S - s0;
    const int n = (int)(rem < 256 ? rem : 256);
    T::load(raw, s0, n, tile);
    __syncthreads();
    if ((int)threadIdx.x < n) {
      const long s = s0 + threadIdx.x;
      const int m = UNI ? __builtin_amdgcn_readfirstlane(frame_of(s0, spf)) : frame_of(s, spf);
      const V3 x = ldv3(xyz + s * 3);
      Blend<B> bl;
      blend_forward<B>(x, aff + (size_t)m * B * 12, tile + threadIdx.x * T::CP, sr + (size_t)m * B * 4, sd + (size_t)m * B * 4, bl);
      const float i = bl.inv;
      const V3 y = dq_apply(bl.rw[0] * i, V3{bl.rw[1], bl.rw[2], bl.rw[3]} * i, bl.dw4[0] * i, V3{bl.dw4[1], bl.dw4[2], bl.dw4[3]} * i, x);
      stv3(out + s * 3, y);
      if (ent) ent[s] = bl.lse_minus_max;
      if (dskin) {
        float q = 0.f;
#pragma unroll
        for (int b = 0; b < B; ++b) q += bl.dl[b] * bl.dl[b];
        dskin[s] = q / (float)B;   // warping.py:332
      }
    }
    __syncthreads();
  }
}

// per-sample part of the adjoint.  work = [coef (S,B): p_b*sign_b][gw (S,8): g_rw | g_dw][gsk (S,B)][xx (S,10)]: the first
// pair feeds the per-frame Gram reduction of the se3 gradient, the second pair the one of the bone-coordinate path:
// skin_b = -(|c_b|^2 + delta_b) gives dL/dc_b = gsk_b * c_b with gsk_b = -2 dL/dskin_b -- rank-structured, so instead
// of materialising an (S,3B) gradient it is (a) pushed to the point here (g_xyz += sum_b Abar_b[:, :3]^T (gsk_b c_b)) and
// (b) reduced per frame as second moments Q_b = sum_s gsk_sb [x,1][x,1]^T (c_b is affine in x), from which
// k_bone_gram_from_moments rebuilds the Gram matrix the parameter chain rule needs.
// UNI: spf is a multiple of 256, so a block's 256 samples share ONE frame: the frame index is wave-uniform (readfirstlane) and the
// per-frame tables (25 x (12 + 4 + 4) floats) are fetched with scalar loads instead of being held in ~200 vector registers.
// FUSE (needs UNI: a tile of 256 samples belongs to ONE frame): the two per-frame Gram reductions -- g_se3[m] += coef^T gw (B x 8) and
// Q[m] += gsk^T xx (B x 10) -- are formed HERE from the LDS tiles, so coef / gw / gsk / xx never go to HBM (272 bytes per sample written and
// read back by k_gram_pf_rb before round 3: 48 of the skinning adjoint's ms per step).  Thread (slice, b) walks every `slices`-th row of the
// tile with the 8 (10) outputs of bone b in registers; the slices meet in LDS, the block keeps its sums in LDS across its tiles and
// flushes them with one atomic per output when the frame changes and at the end.
template <int B, bool UNI, bool FUSE, bool ACC = false>
__global__ void __launch_bounds__(256, (FUSE ? 2 : 1)) k_blend_bwd(const float* __restrict__ xyz, const float* __restrict__ aff, const float* __restrict__ raw,
                                                    const float* __restrict__ sr, const float* __restrict__ sd, const float* __restrict__ g_out,
                                                    const float* __restrict__ g_ent, const float* __restrict__ g_dskin, long S, int spf,
                                                    float* __restrict__ g_xyz, float* __restrict__ g_raw, float* __restrict__ work,
                                                    float* __restrict__ g_se3, float* __restrict__ Qm) {
  // ACC: g_xyz and g_raw are ADDED to (the second blend of one skinning field: both adjoints land in one pair of buffers).  A template
  // parameter: as a runtime flag it cost the fused kernel 8 registers (264: one wave per SIMD instead of two, 35 -> 60 ms per step)
  constexpr int accumulate = ACC ? 1 : 0;
  static_assert(!FUSE || UNI, "the fused Gram reduction needs frame-uniform tiles");
  // The (S,B) operands -- raw in; coef, gsk, g_raw out -- go through ONE LDS tile of 256 rows, one after the other (RowTile): the
  // per-thread row accesses of the first version ran this kernel at 1.7 TB/s (a third of what its 516 bytes per sample allow).
  using T = RowTile<B>;
  __shared__ float tile[T::FLOATS];
  constexpr int AUXW = 12;                 // floats per row of the second operand (gw: 8, xx: 10), 16-byte aligned rows
  constexpr int SLICES = 256 / B;          // (slice, bone) threads of the Gram loops
  static_assert(T::FLOATS >= SLICES * B * 10, "the tile doubles as the slice-reduction scratch");
  __shared__ __attribute__((aligned(16))) float aux[FUSE ? 256 * AUXW : 4];
  __shared__ float lacc[FUSE ? B * 18 : 1];  // block sums: [B x 8 | B x 10]
  int cur_m = -1;
  if (FUSE) {
    for (int e = threadIdx.x; e < B * 18; e += 256) lacc[e] = 0.f;
    // (the first __syncthreads of the loop orders this against the first use)
  }
  auto flush = [&](int m) {  // block sums -> the per-frame accumulators; callers sync before and after
    for (int e = threadIdx.x; e < B * 18; e += 256) {
      const float v = lacc[e];
      if (v != 0.f) atomicAdd(e < B * 8 ? g_se3 + (size_t)m * B * 8 + e : Qm + (size_t)m * B * 10 + (e - B * 8), v);
      lacc[e] = 0.f;
    }
  };
  // acc[j] = sum over the rows k = sl, sl + SLICES, ... < n of tile[k][i] * aux[k][j]; then the slices are summed through the tile
  auto gram = [&](int n, int ncol, int off) {
    const int t = threadIdx.x, sl = t / B, i = t - sl * B;
    const bool act = sl < SLICES;
    float acc[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) acc[j] = 0.f;
    if (act)
      for (int k = sl; k < n; k += SLICES) {
        const float a = tile[k * T::CP + i];
        const float4 b0 = *reinterpret_cast<const float4*>(aux + k * AUXW), b1 = *reinterpret_cast<const float4*>(aux + k * AUXW + 4);
        acc[0] += a * b0.x; acc[1] += a * b0.y; acc[2] += a * b0.z; acc[3] += a * b0.w;
        acc[4] += a * b1.x; acc[5] += a * b1.y; acc[6] += a * b1.z; acc[7] += a * b1.w;
        if (ncol == 10) {
          const float2 b2 = *reinterpret_cast<const float2*>(aux + k * AUXW + 8);
          acc[8] += a * b2.x; acc[9] += a * b2.y;
        }
      }
    __syncthreads();  // every thread is done with the tile's rows: it becomes the scratch of the slice reduction
    if (act)
      for (int j = 0; j < ncol; ++j) tile[(sl * B + i) * ncol + j] = acc[j];
    __syncthreads();
    for (int e = t; e < B * ncol; e += 256) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < SLICES; ++q) v += tile[q * B * ncol + e];
      lacc[off + e] += v;
    }
    __syncthreads();
  };
  for (long s0 = (long)blockIdx.x * 256; s0 < S; s0 += (long)gridDim.x * 256) {
    const long rem = S - s0;
    const int n = (int)(rem < 256 ? rem : 256);
    const bool active = (int)threadIdx.x < n;
    const long s = s0 + threadIdx.x;
    float* row = tile + threadIdx.x * T::CP;
    if (FUSE) {
      const int m_tile = __builtin_amdgcn_readfirstlane(frame_of(s0, spf));
      if (m_tile != cur_m) {  // block-uniform
        if (cur_m >= 0) { __syncthreads(); flush(cur_m); }
        cur_m = m_tile;
      }
    }
    T::load(raw, s0, n, tile);
    __syncthreads();
    Blend<B> bl;
    float gp[B];
    float pg = 0.f, ge = 0.f, gds = 0.f;
    unsigned rawpos = 0;  // bit b: raw[s][b] > 0 (the ReLU of the delta-skin head, skinning.py:119)
    const float* affm = aff;
    V3 x = {0, 0, 0};
    if (active) {
      const int m = UNI ? __builtin_amdgcn_readfirstlane(frame_of(s0, spf)) : frame_of(s, spf);
      const float* srm = sr + (size_t)m * B * 4;
      const float* sdm = sd + (size_t)m * B * 4;
      affm = aff + (size_t)m * B * 12;
      x = ldv3(xyz + s * 3);
      const V3 g = ldv3(g_out + s * 3);
#pragma unroll
      for (int b_ = 0; b_ < B; ++b_) rawpos |= (row[b_] > 0.f ? 1u : 0u) << b_;
      blend_forward<B>(x, affm, row, srm, sdm, bl);
      const float i = bl.inv;
      const float w = bl.rw[0] * i, dw = bl.dw4[0] * i;
      const V3 v = V3{bl.rw[1], bl.rw[2], bl.rw[3]} * i, dv = V3{bl.dw4[1], bl.dw4[2], bl.dw4[3]} * i;
      V3 gx = qrot_t(w, v, g);
      float gqw; V3 gqv;
      qrot_gq(w, v, x, g, gqw, gqv);
      // t = 2 (-dw v + w dv + v x dv)
      const float gn_w = gqw + 2.f * dot(dv, g);
      const V3 gn_v = gqv + g * (-2.f * dw) + cross(dv, g) * 2.f;
      const float gd_w = -2.f * dot(v, g);
      const V3 gd_v = g * (2.f * w) + cross(g, v) * 2.f;
      // normalisation: qn = rw * inv, dn = dw4 * inv, inv = 1/|rw|
      const float qg = w * gn_w + dot(v, gn_v);           // qn . g_qn
      const float dg = dw * gd_w + dot(dv, gd_v);          // dn . g_dn
      float grw[4], gdw[4];
      grw[0] = i * (gn_w - w * qg) - w * i * dg;
      grw[1] = i * (gn_v.x - v.x * qg) - v.x * i * dg;
      grw[2] = i * (gn_v.y - v.y * qg) - v.y * i * dg;
      grw[3] = i * (gn_v.z - v.z * qg) - v.z * i * dg;
      gdw[0] = i * gd_w; gdw[1] = i * gd_v.x; gdw[2] = i * gd_v.y; gdw[3] = i * gd_v.z;
      // [g_rw (4) | g_dw (4)]: 32 contiguous bytes per sample, in the work array -- or, fused, in the LDS operand tile of the Gram loop
      float4* wg = FUSE ? reinterpret_cast<float4*>(aux + threadIdx.x * AUXW) : reinterpret_cast<float4*>(work + S * B + s * 8);
      wg[0] = make_float4(grw[0], grw[1], grw[2], grw[3]);
      wg[1] = make_float4(gdw[0], gdw[1], gdw[2], gdw[3]);
      // softmax / entropy / delta adjoint
#pragma unroll
      for (int b_ = 0; b_ < B; ++b_) {
        const float* r = srm + 4 * b_;
        const float* d = sdm + 4 * b_;
        gp[b_] = bl.sg[b_] * (r[0] * grw[0] + r[1] * grw[1] + r[2] * grw[2] + r[3] * grw[3] + d[0] * gdw[0] + d[1] * gdw[1] + d[2] * gdw[2] + d[3] * gdw[3]);
        pg += bl.p[b_] * gp[b_];
      }
      ge = g_ent ? g_ent[s] : 0.f;
      gds = g_dskin ? g_dskin[s] * (2.f / (float)B) : 0.f;
      // dL/dx through c_b = Abar_b [x,1]:  Abar_b[:, :3]^T (gsk * c_b)
#pragma unroll
      for (int b_ = 0; b_ < B; ++b_) {
        const float gskin = bl.p[b_] * (gp[b_] - pg) + ge * (bl.p[b_] - (b_ == bl.anchor ? 1.f : 0.f));
        const float gsk = -2.f * gskin;
        const Aff A = ld_aff(affm + 12 * b_);
        const V3 c = bone_coord(A, x);
        const float u0 = gsk * c.x, u1 = gsk * c.y, u2 = gsk * c.z;
        gx = gx + V3{A.r0.x * u0 + A.r1.x * u1 + A.r2.x * u2, A.r0.y * u0 + A.r1.y * u1 + A.r2.y * u2, A.r0.z * u0 + A.r1.z * u1 + A.r2.z * u2};
      }
      if (accumulate) gx = gx + ldv3(g_xyz + s * 3);
      stv3(g_xyz + s * 3, gx);
      if (!FUSE) {
        float* xx = work + S * (2 * B + 8) + s * 10;  // upper triangle of [x,1][x,1]^T, row-major
        xx[0] = x.x * x.x; xx[1] = x.x * x.y; xx[2] = x.x * x.z; xx[3] = x.x;
        xx[4] = x.y * x.y; xx[5] = x.y * x.z; xx[6] = x.y;
        xx[7] = x.z * x.z; xx[8] = x.z; xx[9] = 1.f;
      }
    }
    // the three (S,B) outputs, one pass each through the tile (values re-derived from the registers of the pass above)
    // skin_b = -(dist2_b + delta_b);  entropy = lse(skin) - max(skin)
    __syncthreads();  // every thread has read its raw row
    if (active) {
#pragma unroll
      for (int b_ = 0; b_ < B; ++b_) row[b_] = bl.p[b_] * bl.sg[b_];  // coef[s][b] = p_b * sign_b
    }
    __syncthreads();
    if (FUSE) gram(n, 8, 0);  // g_se3 += coef^T gw   (ends with a barrier)
    else { T::store(work, s0, n, tile); __syncthreads(); }
    if (active) {
#pragma unroll
      for (int b_ = 0; b_ < B; ++b_) {
        const float gskin = bl.p[b_] * (gp[b_] - pg) + ge * (bl.p[b_] - (b_ == bl.anchor ? 1.f : 0.f));
        row[b_] = -2.f * gskin;  // gsk[s][b]
      }
      if (FUSE) {  // upper triangle of [x,1][x,1]^T, row-major: the second operand of Q += gsk^T xx
        float* xx = aux + threadIdx.x * AUXW;
        xx[0] = x.x * x.x; xx[1] = x.x * x.y; xx[2] = x.x * x.z; xx[3] = x.x;
        xx[4] = x.y * x.y; xx[5] = x.y * x.z; xx[6] = x.y;
        xx[7] = x.z * x.z; xx[8] = x.z; xx[9] = 1.f;
      }
    }
    __syncthreads();
    if (FUSE) gram(n, 10, B * 8);
    else { T::store(work + S * (B + 8), s0, n, tile); __syncthreads(); }
    if (active) {
#pragma unroll
      for (int b_ = 0; b_ < B; ++b_) {
        const float gskin = bl.p[b_] * (gp[b_] - pg) + ge * (bl.p[b_] - (b_ == bl.anchor ? 1.f : 0.f));
        const float gdelta = -gskin + gds * bl.dl[b_];
        row[b_] = ((rawpos >> b_) & 1u) ? 0.1f * gdelta : 0.f;
      }
    }
    __syncthreads();
    T::store(g_raw, s0, n, tile, accumulate);
    __syncthreads();
  }
  if (FUSE && cur_m >= 0) flush(cur_m);
}

// Gram matrix of the blend's bone-coordinate path from its per-frame second moments (see k_blend_bwd):
//   G[m,b,k,j] = sum_s gsk_sb c_sbk [x_s,1]_j = (1/gauss_bk) sum_i A_b[k][i] Q_b[i][j],  A_b = [R_b | t_b],  Q_b symmetric 4x4.
// One thread per (m,b); G (M,B,3,4) is written.
__global__ void __launch_bounds__(64) k_bone_gram_from_moments(const float* __restrict__ ar, const float* __restrict__ ad,
                                                                const float* __restrict__ gauss, const float* __restrict__ Q, int M, int B,
                                                                float* __restrict__ G) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * B) return;
  const int b = i % B;
  const float* r = ar + (size_t)i * 4;
  const float* d = ad + (size_t)i * 4;
  const float w = r[0], dw = d[0];
  const V3 v = ldv3(r + 1), dv = ldv3(d + 1), gs = ldv3(gauss + 3 * b);
  const V3 e0 = {1.f, 0.f, 0.f}, e1 = {0.f, 1.f, 0.f}, e2 = {0.f, 0.f, 1.f};
  const V3 t = (v * dw - dv * w + cross(v, dv)) * 2.f;
  const V3 c0 = qrot(w, v * -1.f, e0), c1 = qrot(w, v * -1.f, e1), c2 = qrot(w, v * -1.f, e2);  // columns of R
  const float A[3][4] = {{c0.x, c1.x, c2.x, t.x}, {c0.y, c1.y, c2.y, t.y}, {c0.z, c1.z, c2.z, t.z}};
  const float* q = Q + (size_t)i * 10;
  const float Qm[4][4] = {{q[0], q[1], q[2], q[3]}, {q[1], q[4], q[5], q[6]}, {q[2], q[5], q[7], q[8]}, {q[3], q[6], q[8], q[9]}};
  const float ig[3] = {1.f / gs.x, 1.f / gs.y, 1.f / gs.z};
  float* o = G + (size_t)i * 12;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) o[4 * k + j] = ig[k] * (A[k][0] * Qm[0][j] + A[k][1] * Qm[1][j] + A[k][2] * Qm[2][j] + A[k][3] * Qm[3][j]);
}

// ---------------------------------------------------------------------------------------------
// per-frame "skinny Gram" reduction:  out[m][i][j] += sum_{s in frame m} A[s][i] * Bm[s][j]
//   A: (S, CA) row-major, CA <= 80 ; Bm: (S, CB) row-major, CB <= 16 ; CA*CB <= 640 (one thread per output).
// Used for the per-frame parameter gradients of the skinning warp: every per-(sample,bone) gradient is linear in
// a handful of per-sample vectors, so the whole reduction over the samples of a frame is one tall-skinny product.
// Block = (frame, 1024-sample chunk): 128-sample tiles of A and Bm are staged through LDS with coalesced loads,
// thread (i,j) accumulates its output over the tile from LDS (conflict-free: consecutive i -> consecutive banks).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(640) k_gram_pf(const float* __restrict__ A, int CA, const float* __restrict__ Bm, int CB, long S, int spf,
                                                  int chunk, float* __restrict__ out) {
  constexpr int TS = 128;
  extern __shared__ float sm[];  // TS*CA + TS*CB
  float* sa = sm;
  float* sb = sm + TS * CA;
  const int m = blockIdx.y;
  const long f0 = (long)m * spf, f1 = min(S, f0 + spf);
  const long c0 = f0 + (long)blockIdx.x * chunk, c1 = min(f1, c0 + chunk);
  if (c0 >= c1) return;
  const int t = threadIdx.x;
  const int i = t / CB, j = t - i * CB;
  const bool active = t < CA * CB;
  float acc = 0.f;
  for (long s0 = c0; s0 < c1; s0 += TS) {
    const int n = (int)min((long)TS, c1 - s0);
    for (int e = t; e < n * CA; e += blockDim.x) sa[e] = A[s0 * CA + e];
    for (int e = t; e < n * CB; e += blockDim.x) sb[e] = Bm[s0 * CB + e];
    __syncthreads();
    if (active)
      for (int k = 0; k < n; ++k) acc += sa[k * CA + i] * sb[k * CB + j];
    __syncthreads();
  }
  if (active) atomicAdd(out + ((size_t)m * CA + i) * CB + j, acc);
}

// Register-blocked variant for the column counts the skinning adjoints use (CB = 4, 8, 10): thread (i, slice) keeps the
// CB outputs of row i in registers and walks every `slices`-th sample of the staged tile, so one A value read from LDS
// feeds CB FMAs (the B values are a broadcast read) -- (1 + CB)/CB LDS reads per FMA instead of 2.
template <int CB>
__global__ void __launch_bounds__(256) k_gram_pf_rb(const float* __restrict__ A, int CA, const float* __restrict__ Bm, long S, int spf, int chunk,
                                                     float* __restrict__ out) {
  constexpr int TS = 128;
  extern __shared__ float sm[];  // TS*CA + TS*CB
  float* sa = sm;
  float* sb = sm + TS * CA;
  const int m = blockIdx.y;
  const long f0 = (long)m * spf, f1 = min(S, f0 + spf);
  const long c0 = f0 + (long)blockIdx.x * chunk, c1 = min(f1, c0 + chunk);
  if (c0 >= c1) return;
  const int t = threadIdx.x;
  const int slices = 256 / CA;
  const int sl = t / CA, i = t - sl * CA;
  const bool active = sl < slices;
  float acc[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) acc[j] = 0.f;
  for (long s0 = c0; s0 < c1; s0 += TS) {
    const int n = (int)min((long)TS, c1 - s0);
    for (int e = t; e < n * CA; e += 256) sa[e] = A[s0 * CA + e];
    for (int e = t; e < n * CB; e += 256) sb[e] = Bm[s0 * CB + e];
    __syncthreads();
    if (active)
      for (int k = sl; k < n; k += slices) {
        const float a = sa[k * CA + i];
#pragma unroll
        for (int j = 0; j < CB; ++j) acc[j] += a * sb[k * CB + j];
      }
    __syncthreads();
  }
  // reduce the slices inside the block (the staging buffer is free now): one atomic per output and block -- per-thread
  // atomics on the M*CA*CB output words serialise in L2 (measured: 10x more same-address atomics cost 0.6 s per step)
  if (active) {
#pragma unroll
    for (int j = 0; j < CB; ++j) sm[(sl * CA + i) * CB + j] = acc[j];
  }
  __syncthreads();
  for (int e = t; e < CA * CB; e += 256) {
    float v = 0.f;
    for (int q = 0; q < slices; ++q) v += sm[q * CA * CB + e];
    atomicAdd(out + (size_t)m * CA * CB + e, v);
  }
}

}  // namespace lab4d
using namespace lab4d;

#define SKIN_DISPATCH(B, ...)                                                        \
  switch (B) {                                                                        \
    case 25: { constexpr int NB = 25; __VA_ARGS__; break; }                           \
    case 18: { constexpr int NB = 18; __VA_ARGS__; break; }                           \
    default: set_error("skinning kernels are instantiated for B = 25 (bob, skel-quad) and 18 (skel-human); got %d", B); \
             return LAB4D_EINVAL;                                                     \
  }

static inline int sgrid(long n) { int g = div_up(n, 256); return g > 8192 ? 8192 : (g < 1 ? 1 : g); }

extern "C" int lab4d_bone_coords_forward(const float* xyz, const float* ar, const float* ad, const float* gauss, int S, int spf, int M, int B,
                                         float* out, void* stream) {
  LAB4D_REQUIRE(xyz && ar && ad && gauss && out, "bone_coords_forward: null pointer");
  LAB4D_REQUIRE(spf > 0 && (long)M * spf >= S, "bone_coords_forward: M*spf < S");
  if (S == 0) return LAB4D_OK;
  if (spf % 256 == 0) { SKIN_DISPATCH(B, hipLaunchKernelGGL((k_bone_fwd<NB, true>), dim3(sgrid(S)), dim3(256), 0, (hipStream_t)stream, xyz, ar, ad, gauss, (long)S, spf, out)); }
  else { SKIN_DISPATCH(B, hipLaunchKernelGGL((k_bone_fwd<NB, false>), dim3(sgrid(S)), dim3(256), 0, (hipStream_t)stream, xyz, ar, ad, gauss, (long)S, spf, out)); }
  return check_launch("bone_coords_forward");
}

extern "C" int lab4d_bone_coords_backward(const float* xyz, const float* ar, const float* ad, const float* gauss, const float* g_bone, int S,
                                          int spf, int M, int B, float* g_xyz, float* g_ar, float* g_ad, float* g_gauss, void* stream) {
  LAB4D_REQUIRE(xyz && ar && ad && gauss && g_bone, "bone_coords_backward: null pointer");
  LAB4D_REQUIRE(spf > 0 && (long)M * spf >= S, "bone_coords_backward: M*spf < S");
  if (S == 0) return LAB4D_OK;
  hipStream_t st = (hipStream_t)stream;
  if (g_xyz) {
    if (spf % 256 == 0) { SKIN_DISPATCH(B, hipLaunchKernelGGL((k_bone_bwd_x<NB, true, false>), dim3(sgrid(S)), dim3(256), 0, st, ar, gauss, g_bone, (long)S, spf, g_xyz, nullptr, nullptr)); }
    else { SKIN_DISPATCH(B, hipLaunchKernelGGL((k_bone_bwd_x<NB, false, false>), dim3(sgrid(S)), dim3(256), 0, st, ar, gauss, g_bone, (long)S, spf, g_xyz, nullptr, nullptr)); }
  }
  if (g_ar || g_ad || g_gauss) {
    LAB4D_REQUIRE(g_ar && g_ad && g_gauss, "bone_coords_backward: parameter gradients must be requested together");
    const int chunk = 8192;
    const dim3 grid(div_up(spf, chunk), M);
    SKIN_DISPATCH(B, hipLaunchKernelGGL((k_bone_bwd_p<NB>), grid, dim3(256), 0, st, xyz, ar, ad, gauss, g_bone, (long)S, spf, chunk, g_ar, g_ad, g_gauss));
  }
  return check_launch("bone_coords_backward");
}

extern "C" int lab4d_bone_coords_backward_gram(const float* xyz, const float* ar, const float* gauss, const float* g_bone, int S, int spf, int M, int B,
                                               float* g_xyz, float* G, void* stream) {
  LAB4D_REQUIRE(xyz && ar && gauss && g_bone && g_xyz && G, "bone_coords_backward_gram: null pointer");
  LAB4D_REQUIRE(spf > 0 && spf % 256 == 0 && (long)M * spf >= S, "bone_coords_backward_gram: needs spf %% 256 == 0 and M*spf >= S (got spf=%d)", spf);
  hipStream_t st = (hipStream_t)stream;
  if (int e = zero_async(G, (size_t)M * 3 * B * 4 * sizeof(float), st)) return e;
  if (S == 0) return LAB4D_OK;
  int grid = sgrid(S); if (grid > 2048) grid = 2048;  // resident blocks: every block ends with 3B x 4 atomics onto the same M x 3B x 4 words
  SKIN_DISPATCH(B, hipLaunchKernelGGL((k_bone_bwd_x<NB, true, true>), dim3(grid), dim3(256), 0, st, ar, gauss, g_bone, (long)S, spf, g_xyz, xyz, G));
  return check_launch("bone_coords_backward_gram");
}

extern "C" int lab4d_bone_params_from_gram(const float* art_r, const float* art_d, const float* gauss, const float* G, int M, int B,
                                           float* g_art_r, float* g_art_d, float* g_gauss, void* stream) {
  LAB4D_REQUIRE(art_r && art_d && gauss && G, "bone_params_from_gram: null pointer");
  if (M * B == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_bone_param_from_gram, dim3(div_up(M * B, 64)), dim3(64), 0, (hipStream_t)stream, art_r, art_d, gauss, G, M, B, g_art_r,
                     g_art_d, g_gauss);
  return check_launch("bone_params_from_gram");
}

extern "C" int lab4d_bone_affine(const float* art_r, const float* art_d, const float* gauss, int M, int B, float* aff, void* stream) {
  LAB4D_REQUIRE(art_r && art_d && gauss && aff, "bone_affine: null pointer");
  if (M * B == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_bone_affine, dim3(div_up(M * B, 64)), dim3(64), 0, (hipStream_t)stream, art_r, art_d, gauss, M, B, aff);
  return check_launch("bone_affine");
}

extern "C" int lab4d_skin_blend_forward(const float* xyz, const float* art_r, const float* art_d, const float* gauss, const float* raw,
                                        const float* sr, const float* sd, int S, int spf, int M, int B, float* out, float* ent, float* dskin,
                                        float* work, void* stream) {
  LAB4D_REQUIRE(xyz && art_r && art_d && gauss && raw && sr && sd && out && work, "skin_blend_forward: null pointer");
  LAB4D_REQUIRE(spf > 0 && (long)M * spf >= S, "skin_blend_forward: M*spf < S");
  if (S == 0) return LAB4D_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_bone_affine, dim3(div_up(M * B, 64)), dim3(64), 0, st, art_r, art_d, gauss, M, B, work);
  if (spf % 256 == 0) { SKIN_DISPATCH(B, hipLaunchKernelGGL((k_blend_fwd<NB, true>), dim3(sgrid(S)), dim3(256), 0, st, xyz, work, raw, sr, sd, (long)S, spf, out, ent, dskin)); }
  else { SKIN_DISPATCH(B, hipLaunchKernelGGL((k_blend_fwd<NB, false>), dim3(sgrid(S)), dim3(256), 0, st, xyz, work, raw, sr, sd, (long)S, spf, out, ent, dskin)); }
  return check_launch("skin_blend_forward");
}

extern "C" int lab4d_skin_blend_backward_acc(const float* xyz, const float* art_r, const float* art_d, const float* gauss, const float* raw,
                                             const float* sr, const float* sd, const float* g_out, const float* g_ent, const float* g_dskin, int S,
                                             int spf, int M, int B, float* g_xyz, float* g_raw, float* g_se3, float* g_art_r, float* g_art_d,
                                             float* g_gauss, float* work, int accumulate, void* stream);
static bool blend_bwd_fused(int spf, bool has_g_se3) {
  static const int fuse_env = getenv("LAB4D_BLEND_FUSE") ? atoi(getenv("LAB4D_BLEND_FUSE")) : 1;  // 0: the round-2 path (A/B measurements)
  return fuse_env && spf % 256 == 0 && has_g_se3;
}
extern "C" long long lab4d_skin_blend_backward_workspace_floats(int S, int spf, int M, int B, int has_g_se3) {
  return (long long)M * B * 34 + (blend_bwd_fused(spf, has_g_se3 != 0) ? 0 : (long long)S * (2 * B + 18));
}
extern "C" int lab4d_skin_blend_backward(const float* xyz, const float* art_r, const float* art_d, const float* gauss, const float* raw,
                                         const float* sr, const float* sd, const float* g_out, const float* g_ent, const float* g_dskin, int S,
                                         int spf, int M, int B, float* g_xyz, float* g_raw, float* g_se3, float* g_art_r, float* g_art_d,
                                         float* g_gauss, float* work, void* stream) {
  return lab4d_skin_blend_backward_acc(xyz, art_r, art_d, gauss, raw, sr, sd, g_out, g_ent, g_dskin, S, spf, M, B, g_xyz, g_raw, g_se3, g_art_r, g_art_d,
                                       g_gauss, work, 0, stream);
}
extern "C" int lab4d_skin_blend_backward_acc(const float* xyz, const float* art_r, const float* art_d, const float* gauss, const float* raw,
                                             const float* sr, const float* sd, const float* g_out, const float* g_ent, const float* g_dskin, int S,
                                             int spf, int M, int B, float* g_xyz, float* g_raw, float* g_se3, float* g_art_r, float* g_art_d,
                                             float* g_gauss, float* work, int accumulate, void* stream) {
  LAB4D_REQUIRE(xyz && art_r && art_d && gauss && raw && sr && sd && g_out && g_xyz && g_raw && work, "skin_blend_backward: null pointer");
  LAB4D_REQUIRE(spf > 0 && (long)M * spf >= S, "skin_blend_backward: M*spf < S");
  if (S == 0) return LAB4D_OK;
  hipStream_t st = (hipStream_t)stream;
  // work = [aff (M,B,12) | Q (M,B,10) | G (M,B,12)] and, for the unfused path only, behind them [coef (S,B) | gw (S,8) | gsk (S,B) | xx (S,10)];
  // aff first: its rows are read as float4 and the caller's buffer is 16-byte aligned
  float* aff = work;
  float* Q = work + (size_t)M * B * 12;   // (M,B,10), zero-filled here
  float* G = Q + (size_t)M * B * 10;      // (M,B,3,4)
  float* ws = G + (size_t)M * B * 12;
  const bool fused = blend_bwd_fused(spf, g_se3 != nullptr);
  const bool params = g_art_r || g_art_d || g_gauss;
  hipLaunchKernelGGL(k_bone_affine, dim3(div_up(M * B, 64)), dim3(64), 0, st, art_r, art_d, gauss, M, B, aff);
  if (fused || params)
    if (int e = zero_async(Q, (size_t)M * B * 10 * sizeof(float), st)) return e;
  if (fused) {
    // one resident set of blocks walking the samples (every block ends with B x 18 atomics onto the same M x B x 18 words)
    int grid = sgrid(S); if (grid > 2048) grid = 2048;
#define BLEND_BWD(UNI_, FUSE_, GRID_)                                                                                                              \
  do {                                                                                                                                              \
    if (accumulate) { SKIN_DISPATCH(B, hipLaunchKernelGGL((k_blend_bwd<NB, UNI_, FUSE_, true>), dim3(GRID_), dim3(256), 0, st, xyz, aff, raw, sr, sd, g_out, \
                                                          g_ent, g_dskin, (long)S, spf, g_xyz, g_raw, ws, g_se3, Q)); }                              \
    else { SKIN_DISPATCH(B, hipLaunchKernelGGL((k_blend_bwd<NB, UNI_, FUSE_, false>), dim3(GRID_), dim3(256), 0, st, xyz, aff, raw, sr, sd, g_out,       \
                                               g_ent, g_dskin, (long)S, spf, g_xyz, g_raw, ws, g_se3, Q)); }                                         \
  } while (0)
    BLEND_BWD(true, true, grid);
  } else if (spf % 256 == 0) {
    BLEND_BWD(true, false, sgrid(S));
  } else {
    BLEND_BWD(false, false, sgrid(S));
  }
#undef BLEND_BWD
  if (int e = check_launch("skin_blend_backward")) return e;
  if (g_se3 && !fused)
    if (int e = lab4d_gram_per_frame(ws, B, ws + (size_t)S * B, 8, S, spf, M, g_se3, stream)) return e;
  if (params) {
    // bone-coordinate path: per-frame moments -> Gram matrix -> (M,B)-sized chain rule
    if (!fused)
      if (int e = lab4d_gram_per_frame(ws + (size_t)S * (B + 8), B, ws + (size_t)S * (2 * B + 8), 10, S, spf, M, Q, stream)) return e;
    hipLaunchKernelGGL(k_bone_gram_from_moments, dim3(div_up(M * B, 64)), dim3(64), 0, st, art_r, art_d, gauss, Q, M, B, G);
    if (int e = check_launch("bone_gram_from_moments")) return e;
    return lab4d_bone_params_from_gram(art_r, art_d, gauss, G, M, B, g_art_r, g_art_d, g_gauss, stream);
  }
  return LAB4D_OK;
}

// ---------------------------------------------------------------------------------------------
// gaussian-bone density: max_b exp(-0.5 |x - c_b|^2 / 0.01^2) * ibeta
//   (deformable.py:329-356, warping.py:355-387, transforms.py:28-40; centres = frame-0 rest bones)
// ---------------------------------------------------------------------------------------------
namespace lab4d {
__global__ void __launch_bounds__(256) k_gauss_density_fwd(const float* __restrict__ xyz, const float* __restrict__ centres, int B,
                                                            const float* __restrict__ ibeta_p, long S, float* __restrict__ out, int* __restrict__ best) {
  const float ibeta = *ibeta_p;
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (long)gridDim.x * blockDim.x) {
    const V3 x = ldv3(xyz + s * 3);
    float dmin = INFINITY;
    int bi = 0;
    for (int b = 0; b < B; ++b) {
      const V3 d = x - ldv3(centres + 3 * b);
      const float d2 = dot(d, d);
      if (d2 < dmin) { dmin = d2; bi = b; }
    }
    out[s] = expf(-0.5f * (dmin / (0.01f * 0.01f))) * ibeta;
    if (best) best[s] = bi;
  }
}
// g_xyz (S,3) written; g_centres (B,3) and g_ibeta (1) accumulated
__global__ void __launch_bounds__(256) k_gauss_density_bwd(const float* __restrict__ xyz, const float* __restrict__ centres, int B,
                                                            const float* __restrict__ ibeta_p, const int* __restrict__ best, const float* __restrict__ g, long S,
                                                            float* __restrict__ g_xyz, float* __restrict__ g_centres, float* __restrict__ g_ibeta) {
  extern __shared__ float acc[];  // 3B + 1
  const float ibeta = *ibeta_p;
  for (int i = threadIdx.x; i < 3 * B + 1; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (long)gridDim.x * blockDim.x) {
    const int b = best[s];
    const V3 d = ldv3(xyz + s * 3) - ldv3(centres + 3 * b);
    const float e = expf(-0.5f * (dot(d, d) / (0.01f * 0.01f)));
    const float gs = g[s];
    const V3 gx = d * (-gs * e * ibeta / (0.01f * 0.01f));
    if (g_xyz) stv3(g_xyz + s * 3, gx);
    atomicAdd(&acc[3 * b + 0], -gx.x); atomicAdd(&acc[3 * b + 1], -gx.y); atomicAdd(&acc[3 * b + 2], -gx.z);
    atomicAdd(&acc[3 * B], gs * e);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * B; i += blockDim.x) if (acc[i] != 0.f) atomicAdd(g_centres + i, acc[i]);
  if (threadIdx.x == 0 && g_ibeta) atomicAdd(g_ibeta, acc[3 * B]);
}
}  // namespace lab4d

extern "C" int lab4d_gauss_density_forward(const float* xyz, const float* centres, int B, const float* ibeta, int S, float* out, int* best, void* stream) {
  LAB4D_REQUIRE(xyz && centres && out && ibeta, "gauss_density_forward: null pointer");
  if (S == 0) return LAB4D_OK;
  hipLaunchKernelGGL(lab4d::k_gauss_density_fwd, dim3(sgrid(S)), dim3(256), 0, (hipStream_t)stream, xyz, centres, B, ibeta, (long)S, out, best);
  return check_launch("gauss_density_forward");
}
extern "C" int lab4d_gauss_density_backward(const float* xyz, const float* centres, int B, const float* ibeta, const int* best, const float* g, int S,
                                            float* g_xyz, float* g_centres, float* g_ibeta, void* stream) {
  LAB4D_REQUIRE(xyz && centres && best && g && g_centres, "gauss_density_backward: null pointer");
  LAB4D_REQUIRE(B <= 64, "gauss_density_backward: B too large");
  if (S == 0) return LAB4D_OK;
  int grid = sgrid(S); if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(lab4d::k_gauss_density_bwd, dim3(grid), dim3(256), (3 * B + 1) * sizeof(float), (hipStream_t)stream, xyz, centres, B, ibeta, best, g,
                     (long)S, g_xyz, g_centres, g_ibeta);
  return check_launch("gauss_density_backward");
}

extern "C" int lab4d_gram_per_frame(const float* A, int CA, const float* Bm, int CB, int S, int spf, int M, float* out, void* stream) {
  LAB4D_REQUIRE(A && Bm && out, "gram_per_frame: null pointer");
  LAB4D_REQUIRE(CA >= 1 && CA <= 80 && CB >= 1 && CB <= 16 && CA * CB <= 640, "gram_per_frame: need CA <= 80, CB <= 16, CA*CB <= 640 (got %d, %d)", CA, CB);
  LAB4D_REQUIRE(spf > 0 && (long)M * spf >= S, "gram_per_frame: M*spf < S");
  if (S == 0) return LAB4D_OK;
  const int chunk = 1024;
  const dim3 grid(div_up(spf, chunk), M);
  const size_t lds = 128 * (CA + CB) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (CB == 4) hipLaunchKernelGGL(lab4d::k_gram_pf_rb<4>, grid, dim3(256), lds, st, A, CA, Bm, (long)S, spf, chunk, out);
  else if (CB == 8) hipLaunchKernelGGL(lab4d::k_gram_pf_rb<8>, grid, dim3(256), lds, st, A, CA, Bm, (long)S, spf, chunk, out);
  else if (CB == 10) hipLaunchKernelGGL(lab4d::k_gram_pf_rb<10>, grid, dim3(256), lds, st, A, CA, Bm, (long)S, spf, chunk, out);
  else hipLaunchKernelGGL(lab4d::k_gram_pf, grid, dim3(640), lds, st, A, CA, Bm, CB, (long)S, spf, chunk, out);
  return check_launch("gram_per_frame");
}
