// The per-frame (M-row) MLPs in front of the hot path -- TimeEmbedding, TimeMLP and the heads of CameraMLP / IntrinsicsMLP /
// ArticulationSkelMLP / AppearanceEmbedding -- as a PROGRAM of dense layers over a per-row strip: one launch forward, two backward.
// Contract and the reference lines it replaces: include/lab4d_rowmlp.h.
//
// This is a launch-count problem (M <= a few hundred rows, ~40 torch launches forward and ~100 backward per module and step in the
// reference), not a FLOP or bandwidth one: 256 rows x 10 layers of 256 x 256 are 0.17 GFLOP.  So: plain fp32 FMA (the precision the
// reference computes these modules in; no packing: the weights are read in nn.Linear's own (out, in) layout), a workgroup owns R
// rows for the whole program -- rows are independent, so the layers follow each other behind workgroup barriers inside ONE launch --
// and every global access is coalesced: the forward transposes 32-column tiles of W through LDS, the input-gradient chain reads W
// along its rows, the parameter kernel contracts over the rows with X along its columns.  No atomics: every gradient element has one
// owner, results are deterministic.
#include "common.hpp"

namespace lab4d {

constexpr int RM_R = 8;       // rows per workgroup
constexpr int RM_T = 256;     // threads per workgroup
constexpr int RM_KT = 32;     // k tile of the forward's W transposition
constexpr int RM_MAXD = 1024; // widest layer input / output
constexpr int RM_TO = 8;      // outputs per job of the parameter kernel
constexpr int RM_MC = 64;     // rows per staged chunk of the parameter kernel

__device__ __forceinline__ float rm_tid(const lab4d_rowmlp_prog& p, long f) {
  // embedding.py:177-184 in fp32, operation by operation: (tid_sub - vid_len / 2) / max_ts * 2, then * time_scale
  const float sub = (float)(f - p.vstart[f]);
  const float half = (float)p.vidlen[f] / 2.0f;
  return (sub - half) / p.max_ts * 2.0f * p.time_scale;
}

__global__ void __launch_bounds__(RM_T) k_rowmlp_fwd(lab4d_rowmlp_prog p, float* __restrict__ work, int M) {
  __shared__ float xs[RM_R][RM_MAXD];
  __shared__ float wt[RM_KT][RM_T + 1];
  const int tid = threadIdx.x, r0 = blockIdx.x * RM_R;
  const int nr = min(RM_R, M - r0);
  const size_t rs = (size_t)p.row_stride;
  // ---- time prologue ----
  if (p.frame_id != nullptr) {
    const int nf = 2 * p.n_freq + 1;
    for (int e = tid; e < nr * nf; e += RM_T) {
      const int r = e / nf, c = e - r * nf;
      const float t = rm_tid(p, (long)p.frame_id[r0 + r]);
      float v = t;
      if (c > 0) {
        const float ang = exp2f((float)((c - 1) >> 1)) * t;  // freq_bands[k] * x (embedding.py:97-100)
        v = ((c - 1) & 1) ? cosf(ang) : sinf(ang);
      }
      work[(r0 + r) * rs + p.four_col + c] = v;
    }
    for (int e = tid; e < nr * p.inst_dim; e += RM_T) {
      const int r = e / p.inst_dim, c = e - r * p.inst_dim;
      const long f = (long)p.frame_id[r0 + r];
      const long row = p.inst_rows == 1 ? 0 : (long)p.vid[f];
      work[(r0 + r) * rs + p.inst_col + c] = p.inst_W[row * p.inst_dim + c];
    }
  }
  // ---- external inputs -> the strip ----
  for (int q = 0; q < p.n_in; ++q) {
    const lab4d_rowmlp_io io = p.in[q];
    for (int e = tid; e < nr * io.width; e += RM_T) {
      const int r = e / io.width, c = e - r * io.width;
      work[(r0 + r) * rs + io.col + c] = io.ptr[(size_t)(r0 + r) * io.width + c];
    }
  }
  __syncthreads();
  for (int l = 0; l < p.n_layers; ++l) {
    const lab4d_rowmlp_layer L = p.layer[l];
    // stage the rows' inputs
    for (int e = tid; e < RM_R * L.in_dim; e += RM_T) {
      const int r = e / L.in_dim, k = e - r * L.in_dim;
      xs[r][k] = r < nr ? work[(r0 + r) * rs + L.src_col + k] : 0.f;
    }
    for (int o0 = 0; o0 < L.out_dim; o0 += RM_T) {
      const int o = o0 + tid;
      float acc[RM_R];
      const float bias = (L.b != nullptr && o < L.out_dim) ? L.b[o] : 0.f;
#pragma unroll
      for (int r = 0; r < RM_R; ++r) acc[r] = bias;
      for (int k0 = 0; k0 < L.in_dim; k0 += RM_KT) {
        __syncthreads();  // (xs staged / the previous tile consumed)
        // W[o0 .. o0 + 255][k0 .. k0 + 31] -> wt[k][o] : 32 consecutive lanes read 128 contiguous bytes of one row of W
#pragma unroll 4
        for (int it = 0; it < RM_KT; ++it) {
          const int idx = it * RM_T + tid, ol = idx / RM_KT, kk = idx - ol * RM_KT;
          const int oo = o0 + ol, k = k0 + kk;
          wt[kk][ol] = (oo < L.out_dim && k < L.in_dim) ? L.W[(size_t)oo * L.in_dim + k] : 0.f;
        }
        __syncthreads();
        const int kn = min(RM_KT, L.in_dim - k0);
        for (int kk = 0; kk < kn; ++kk) {
          const float w = wt[kk][tid];
#pragma unroll
          for (int r = 0; r < RM_R; ++r) acc[r] = fmaf(w, xs[r][k0 + kk], acc[r]);
        }
      }
      if (o < L.out_dim) {
#pragma unroll
        for (int r = 0; r < RM_R; ++r)
          if (r < nr) work[(r0 + r) * rs + L.dst_col + o] = L.relu ? fmaxf(acc[r], 0.f) : acc[r];
      }
    }
    __syncthreads();  // this layer's outputs are visible to the workgroup before the next layer stages them
  }
  // ---- the strip -> the caller's output tensors ----
  for (int q = 0; q < p.n_out; ++q) {
    const lab4d_rowmlp_io io = p.out[q];
    if (io.ptr == nullptr) continue;
    for (int e = tid; e < nr * io.width; e += RM_T) {
      const int r = e / io.width, c = e - r * io.width;
      io.ptr[(size_t)(r0 + r) * io.width + c] = work[(r0 + r) * rs + io.col + c];
    }
  }
}

// dZ in place + input gradients accumulated into the sources, layers in reverse
__global__ void __launch_bounds__(RM_T) k_rowmlp_bwd_chain(lab4d_rowmlp_prog p, const float* __restrict__ work, float* __restrict__ gwork, int M) {
  __shared__ float dzs[RM_R][RM_MAXD];
  const int tid = threadIdx.x, r0 = blockIdx.x * RM_R;
  const int nr = min(RM_R, M - r0);
  const size_t rs = (size_t)p.row_stride;
  // ---- dL/d(strip) of this workgroup's rows: zero, then the gradients of the outputs the caller consumed ----
  for (int e = tid; e < nr * p.row_stride; e += RM_T) {
    const int r = e / p.row_stride, c = e - r * p.row_stride;
    gwork[(r0 + r) * rs + c] = 0.f;
  }
  __syncthreads();
  for (int q = 0; q < p.n_out; ++q) {
    const lab4d_rowmlp_io io = p.out[q];
    if (io.ptr == nullptr) continue;
    for (int e = tid; e < nr * io.width; e += RM_T) {
      const int r = e / io.width, c = e - r * io.width;
      gwork[(r0 + r) * rs + io.col + c] += io.ptr[(size_t)(r0 + r) * io.width + c];  // (+=: two outputs may name overlapping columns)
    }
    __syncthreads();
  }
  for (int l = p.n_layers - 1; l >= 0; --l) {
    const lab4d_rowmlp_layer L = p.layer[l];
    for (int e = tid; e < RM_R * L.out_dim; e += RM_T) {
      const int r = e / L.out_dim, o = e - r * L.out_dim;
      float g = 0.f;
      if (r < nr) {
        const size_t a = (r0 + r) * rs + L.dst_col + o;
        g = gwork[a];
        if (L.relu && !(work[a] > 0.f)) g = 0.f;
        gwork[a] = g;
      }
      dzs[r][o] = g;
    }
    __syncthreads();
    for (int i = tid; i < L.in_dim; i += RM_T) {
      float acc[RM_R];
#pragma unroll
      for (int r = 0; r < RM_R; ++r) acc[r] = 0.f;
      for (int o = 0; o < L.out_dim; ++o) {
        const float w = L.W[(size_t)o * L.in_dim + i];  // consecutive threads: consecutive columns of one row of W
#pragma unroll
        for (int r = 0; r < RM_R; ++r) acc[r] = fmaf(dzs[r][o], w, acc[r]);
      }
#pragma unroll
      for (int r = 0; r < RM_R; ++r)
        if (r < nr) gwork[(r0 + r) * rs + L.src_col + i] += acc[r];
    }
    __syncthreads();
  }
  // ---- gradients of the external inputs ----
  for (int q = 0; q < p.n_in; ++q) {
    const lab4d_rowmlp_io io = p.in[q];
    if (io.ptr == nullptr) continue;
    for (int e = tid; e < nr * io.width; e += RM_T) {
      const int r = e / io.width, c = e - r * io.width;
      io.ptr[(size_t)(r0 + r) * io.width + c] = gwork[(r0 + r) * rs + io.col + c];
    }
  }
}

// jobs: layer l has ceil(out_dim / RM_TO) jobs (dW rows o0 .. o0 + 7 and their db), then inst_rows jobs for d_inst_W
__global__ void __launch_bounds__(RM_T) k_rowmlp_bwd_param(lab4d_rowmlp_prog p, const float* __restrict__ work, const float* __restrict__ gwork, int M) {
  __shared__ float dzt[RM_MC][RM_TO];
  const int tid = threadIdx.x;
  const size_t rs = (size_t)p.row_stride;
  int job = blockIdx.x, l = 0;
  for (; l < p.n_layers; ++l) {
    const int nj = (p.layer[l].out_dim + RM_TO - 1) / RM_TO;
    if (job < nj) break;
    job -= nj;
  }
  if (l == p.n_layers) {
    // d_inst_W[v][c] = sum over the rows of video v (a single-row table takes every row)
    if (p.d_inst_W == nullptr || p.frame_id == nullptr || job >= p.inst_rows) return;
    for (int c = tid; c < p.inst_dim; c += RM_T) {
      float acc = 0.f;
      for (int m = 0; m < M; ++m) {
        const long row = p.inst_rows == 1 ? 0 : (long)p.vid[(long)p.frame_id[m]];
        if (row == job) acc += gwork[m * rs + p.inst_col + c];
      }
      float* dst = p.d_inst_W + (size_t)job * p.inst_dim + c;
      *dst = p.acc_inst ? *dst + acc : acc;
    }
    return;
  }
  const lab4d_rowmlp_layer L = p.layer[l];
  if (L.dW == nullptr && L.db == nullptr) return;
  const int o0 = job * RM_TO, no = min(RM_TO, L.out_dim - o0);
  const int ni = (L.in_dim + RM_T - 1) / RM_T;  // column passes of this thread (<= 4)
  float acc[4][RM_TO];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < RM_TO; ++j) acc[q][j] = 0.f;
  float bsum = 0.f;
  for (int m0 = 0; m0 < M; m0 += RM_MC) {
    const int mc = min(RM_MC, M - m0);
    __syncthreads();
    for (int e = tid; e < RM_MC * RM_TO; e += RM_T) {
      const int mm = e / RM_TO, j = e - mm * RM_TO;
      dzt[mm][j] = (mm < mc && j < no) ? gwork[(m0 + mm) * rs + L.dst_col + o0 + j] : 0.f;
    }
    __syncthreads();
    if (L.dW != nullptr) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = q * RM_T + tid;
        if (q < ni && i < L.in_dim) {
          for (int mm = 0; mm < mc; ++mm) {
            const float x = work[(m0 + mm) * rs + L.src_col + i];
#pragma unroll
            for (int j = 0; j < RM_TO; ++j) acc[q][j] = fmaf(dzt[mm][j], x, acc[q][j]);
          }
        }
      }
    }
    if (tid < RM_TO)
      for (int mm = 0; mm < mc; ++mm) bsum += dzt[mm][tid];
  }
  if (L.dW != nullptr) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = q * RM_T + tid;
      if (q < ni && i < L.in_dim)
#pragma unroll
        for (int j = 0; j < RM_TO; ++j)
          if (j < no) {
            float* dst = L.dW + (size_t)(o0 + j) * L.in_dim + i;
            *dst = (L.acc & 1) ? *dst + acc[q][j] : acc[q][j];
          }
    }
  }
  if (L.db != nullptr && tid < no) L.db[o0 + tid] = (L.acc & 2) ? L.db[o0 + tid] + bsum : bsum;
}

// ---- epilogues of the per-frame modules: the handful of M x 4 element-wise ops behind the heads, one launch each way instead of ~12 / ~25 ----------
__device__ __forceinline__ long row_video(const int64_t* frame_id, const int64_t* vid, int V, int m) {
  return V == 1 ? 0 : (long)vid[(long)frame_id[m]];
}
// CameraMLP.get_vals behind the heads (pose.py:126-147): quat = F.normalize(raw); base = F.normalize(base_quat[video]); out = quaternion_mul(quat, base)
__global__ void __launch_bounds__(256) k_cam_epi_fwd(const float* __restrict__ raw, const float* __restrict__ base, const int64_t* __restrict__ frame_id,
                                                     const int64_t* __restrict__ vid, int M, int V, float* __restrict__ out) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float4 r = *reinterpret_cast<const float4*>(raw + 4 * (size_t)m);
  const float4 b = *reinterpret_cast<const float4*>(base + 4 * row_video(frame_id, vid, V, m));
  const float ir = 1.f / fmaxf(sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w), 1e-12f);  // F.normalize: x / max(|x|, eps)
  const float ib = 1.f / fmaxf(sqrtf(b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w), 1e-12f);
  const float aw = r.x * ir, ax = r.y * ir, ay = r.z * ir, az = r.w * ir, bw = b.x * ib, bx = b.y * ib, by = b.z * ib, bz = b.w * ib;
  *reinterpret_cast<float4*>(out + 4 * (size_t)m) = make_float4(aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                                                                 aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw);
}
// adjoint per row: g_raw (M,4) written; the row's gradient wrt its video's UN-normalised base quaternion -> rowg (M,4) (reduced per video below)
__global__ void __launch_bounds__(256) k_cam_epi_bwd(const float* __restrict__ raw, const float* __restrict__ base, const int64_t* __restrict__ frame_id,
                                                     const int64_t* __restrict__ vid, const float* __restrict__ g_out, int M, int V,
                                                     float* __restrict__ g_raw, float* __restrict__ rowg) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float4 r = *reinterpret_cast<const float4*>(raw + 4 * (size_t)m);
  const float4 b = *reinterpret_cast<const float4*>(base + 4 * row_video(frame_id, vid, V, m));
  const float4 g = *reinterpret_cast<const float4*>(g_out + 4 * (size_t)m);
  const float nr = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w), nb = sqrtf(b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w);
  const float ir = 1.f / fmaxf(nr, 1e-12f), ib = 1.f / fmaxf(nb, 1e-12f);
  const float aw = r.x * ir, ax = r.y * ir, ay = r.z * ir, az = r.w * ir, bw = b.x * ib, bx = b.y * ib, by = b.z * ib, bz = b.w * ib;
  // ga = g (x) conj(b), gb = conj(a) (x) g   (csrc/quaternion.hip k_qmul_bwd)
  float ga[4] = {g.x * bw + g.y * bx + g.z * by + g.w * bz, -g.x * bx + g.y * bw - g.z * bz + g.w * by, -g.x * by + g.y * bz + g.z * bw - g.w * bx,
                 -g.x * bz - g.y * by + g.z * bx + g.w * bw};
  float gb[4] = {g.x * aw + g.y * ax + g.z * ay + g.w * az, -g.x * ax + g.y * aw + g.z * az - g.w * ay, -g.x * ay - g.y * az + g.z * aw + g.w * ax,
                 -g.x * az + g.y * ay - g.z * ax + g.w * aw};
  // y = x / max(|x|, eps): dx = (dy - y (y . dy)) / |x| above the clamp, dy / eps below it
  const float da = aw * ga[0] + ax * ga[1] + ay * ga[2] + az * ga[3], db = bw * gb[0] + bx * gb[1] + by * gb[2] + bz * gb[3];
  const float ka = nr > 1e-12f ? da : 0.f, kb = nb > 1e-12f ? db : 0.f;
  *reinterpret_cast<float4*>(g_raw + 4 * (size_t)m) = make_float4((ga[0] - aw * ka) * ir, (ga[1] - ax * ka) * ir, (ga[2] - ay * ka) * ir, (ga[3] - az * ka) * ir);
  *reinterpret_cast<float4*>(rowg + 4 * (size_t)m) = make_float4((gb[0] - bw * kb) * ib, (gb[1] - bx * kb) * ib, (gb[2] - by * kb) * ib, (gb[3] - bz * kb) * ib);
}
// dst[v][c] (+)= sum over the rows m of video v of rowg[m][c]: one workgroup per video, fixed summation order (deterministic)
__global__ void __launch_bounds__(256) k_rows_to_video(const float* __restrict__ rowg, int C, const int64_t* __restrict__ frame_id, const int64_t* __restrict__ vid,
                                                       int M, int V, float* __restrict__ dst, int acc) {
  __shared__ float part[256];
  const int v = blockIdx.x;
  for (int c = 0; c < C; ++c) {
    float s = 0.f;
    for (int m = threadIdx.x; m < M; m += 256)
      if (row_video(frame_id, vid, V, m) == v) s += rowg[(size_t)m * C + c];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) dst[(size_t)v * C + c] = acc ? dst[(size_t)v * C + c] + part[0] : part[0];
    __syncthreads();
  }
}
// IntrinsicsMLP.get_vals behind the head (intrinsics.py:94-107): f = exp(raw) * exp(base_logfocal[video]); both focal lengths = their mean ("square pixels");
// out = [f_mean, f_mean, ppoint[video]]
__global__ void __launch_bounds__(256) k_intr_epi_fwd(const float* __restrict__ raw, const float* __restrict__ logfocal, const float* __restrict__ ppoint,
                                                      const int64_t* __restrict__ frame_id, const int64_t* __restrict__ vid, int M, int V, float* __restrict__ out) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const long v = row_video(frame_id, vid, V, m);
  const float f0 = expf(raw[2 * (size_t)m]) * expf(logfocal[2 * v]), f1 = expf(raw[2 * (size_t)m + 1]) * expf(logfocal[2 * v + 1]);
  const float fm = (f0 + f1) / 2.f;
  *reinterpret_cast<float4*>(out + 4 * (size_t)m) = make_float4(fm, fm, ppoint[2 * v], ppoint[2 * v + 1]);
}
// adjoint per row: g_raw (M,2) written; rowg (M,4) = [d logfocal (2) | d ppoint (2)] of the row's video
__global__ void __launch_bounds__(256) k_intr_epi_bwd(const float* __restrict__ raw, const float* __restrict__ logfocal, const int64_t* __restrict__ frame_id,
                                                      const int64_t* __restrict__ vid, const float* __restrict__ g_out, int M, int V, float* __restrict__ g_raw,
                                                      float* __restrict__ rowg) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const long v = row_video(frame_id, vid, V, m);
  const float4 g = *reinterpret_cast<const float4*>(g_out + 4 * (size_t)m);
  const float f0 = expf(raw[2 * (size_t)m]) * expf(logfocal[2 * v]), f1 = expf(raw[2 * (size_t)m + 1]) * expf(logfocal[2 * v + 1]);
  const float gf = (g.x + g.y) / 2.f;  // both outputs are the mean of the two
  g_raw[2 * (size_t)m] = gf * f0;
  g_raw[2 * (size_t)m + 1] = gf * f1;
  *reinterpret_cast<float4*>(rowg + 4 * (size_t)m) = make_float4(gf * f0, gf * f1, g.z, g.w);
}

static int rowmlp_check(const lab4d_rowmlp_prog* p, const void* work, int M) {
  LAB4D_REQUIRE(p != nullptr && work != nullptr && M >= 0, "rowmlp: null program / workspace or negative row count");
  LAB4D_REQUIRE(p->n_layers >= 1 && p->n_layers <= LAB4D_ROWMLP_MAX_LAYERS && p->row_stride >= 1, "rowmlp: %d layers (1..%d), row stride %d", p->n_layers,
                LAB4D_ROWMLP_MAX_LAYERS, p->row_stride);
  for (int l = 0; l < p->n_layers; ++l) {
    const lab4d_rowmlp_layer& L = p->layer[l];
    LAB4D_REQUIRE(L.W != nullptr && L.in_dim >= 1 && L.out_dim >= 1 && L.in_dim <= RM_MAXD && L.out_dim <= RM_MAXD, "rowmlp: layer %d: weights missing or %d -> %d outside 1..%d", l,
                  L.in_dim, L.out_dim, RM_MAXD);
    LAB4D_REQUIRE(L.src_col >= 0 && L.dst_col >= 0 && L.src_col + L.in_dim <= p->row_stride && L.dst_col + L.out_dim <= p->row_stride,
                  "rowmlp: layer %d: columns [%d, +%d) -> [%d, +%d) leave the row strip of %d", l, L.src_col, L.in_dim, L.dst_col, L.out_dim, p->row_stride);
    LAB4D_REQUIRE(!(L.dst_col < L.src_col + L.in_dim && L.src_col < L.dst_col + L.out_dim), "rowmlp: layer %d writes into its own input columns", l);
    // every column range is written by ONE producer (the backward turns a layer's output gradient into dZ in place)
    for (int k = 0; k < l; ++k) {
      const lab4d_rowmlp_layer& K = p->layer[k];
      LAB4D_REQUIRE(!(L.dst_col < K.dst_col + K.out_dim && K.dst_col < L.dst_col + L.out_dim), "rowmlp: layers %d and %d write overlapping columns", k, l);
    }
    if (p->frame_id != nullptr) {
      LAB4D_REQUIRE(!(L.dst_col < p->four_col + 2 * p->n_freq + 1 && p->four_col < L.dst_col + L.out_dim), "rowmlp: layer %d writes into the Fourier columns", l);
      LAB4D_REQUIRE(p->inst_dim == 0 || !(L.dst_col < p->inst_col + p->inst_dim && p->inst_col < L.dst_col + L.out_dim), "rowmlp: layer %d writes into the instance-code columns", l);
    }
  }
  LAB4D_REQUIRE(p->n_in >= 0 && p->n_in <= LAB4D_ROWMLP_MAX_IO && p->n_out >= 0 && p->n_out <= LAB4D_ROWMLP_MAX_IO, "rowmlp: %d inputs / %d outputs (0..%d)", p->n_in, p->n_out,
                LAB4D_ROWMLP_MAX_IO);
  for (int q = 0; q < p->n_in + p->n_out; ++q) {
    const lab4d_rowmlp_io& io = q < p->n_in ? p->in[q] : p->out[q - p->n_in];
    LAB4D_REQUIRE(io.col >= 0 && io.width >= 1 && io.col + io.width <= p->row_stride, "rowmlp: %s %d: columns [%d, +%d) leave the row strip of %d", q < p->n_in ? "input" : "output",
                  q < p->n_in ? q : q - p->n_in, io.col, io.width, p->row_stride);
  }
  if (p->frame_id != nullptr) {
    LAB4D_REQUIRE(p->vstart != nullptr && p->vidlen != nullptr && p->n_freq >= 0 && p->n_freq <= 16 && p->max_ts > 0.f, "rowmlp: time prologue: frame tables missing, n_freq %d or max_ts %g",
                  p->n_freq, (double)p->max_ts);
    LAB4D_REQUIRE(p->four_col >= 0 && p->four_col + 2 * p->n_freq + 1 <= p->row_stride, "rowmlp: time prologue: Fourier columns leave the row strip");
    LAB4D_REQUIRE(p->inst_dim >= 0 && (p->inst_dim == 0 || (p->inst_W != nullptr && p->inst_rows >= 1 && p->inst_col >= 0 && p->inst_col + p->inst_dim <= p->row_stride)),
                  "rowmlp: time prologue: instance-code table / columns");
    LAB4D_REQUIRE(!(p->inst_dim > 0 && p->inst_rows > 1 && p->vid == nullptr), "rowmlp: time prologue: a multi-row instance table needs raw_fid_to_vid");
  }
  return LAB4D_OK;
}

}  // namespace lab4d

using namespace lab4d;

extern "C" int lab4d_rowmlp_forward(const lab4d_rowmlp_prog* prog, float* work, int M, void* stream) {
  const int rc = rowmlp_check(prog, work, M);
  if (rc != LAB4D_OK) return rc;
  if (M == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_rowmlp_fwd, dim3((M + RM_R - 1) / RM_R), dim3(RM_T), 0, (hipStream_t)stream, *prog, work, M);
  return check_launch("rowmlp_forward");
}

extern "C" int lab4d_rowmlp_backward(const lab4d_rowmlp_prog* prog, const float* work, float* gwork, int M, void* stream) {
  const int rc = rowmlp_check(prog, work, M);
  if (rc != LAB4D_OK) return rc;
  LAB4D_REQUIRE(gwork != nullptr, "rowmlp_backward: null gradient workspace");
  if (M == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_rowmlp_bwd_chain, dim3((M + RM_R - 1) / RM_R), dim3(RM_T), 0, (hipStream_t)stream, *prog, work, gwork, M);
  int jobs = 0;
  for (int l = 0; l < prog->n_layers; ++l) jobs += (prog->layer[l].out_dim + RM_TO - 1) / RM_TO;
  if (prog->frame_id != nullptr && prog->d_inst_W != nullptr) jobs += prog->inst_rows;
  hipLaunchKernelGGL(k_rowmlp_bwd_param, dim3(jobs), dim3(RM_T), 0, (hipStream_t)stream, *prog, work, (const float*)gwork, M);
  return check_launch("rowmlp_backward");
}

#define EPI_CHECKS(name)                                                                                                   \
  LAB4D_REQUIRE(M >= 0 && V >= 1, name ": bad sizes M=%d V=%d", M, V);                                                   \
  LAB4D_REQUIRE(V == 1 || (frame_id != nullptr && vid != nullptr), name ": several videos need the frame ids and raw_fid_to_vid"); \
  if (M == 0) return LAB4D_OK;

extern "C" int lab4d_camera_epilogue_forward(const float* raw, const float* base_quat, const int64_t* frame_id, const int64_t* vid, int M, int V, float* out,
                                             void* stream) {
  EPI_CHECKS("camera_epilogue_forward");
  LAB4D_REQUIRE(raw && base_quat && out, "camera_epilogue_forward: null pointer");
  hipLaunchKernelGGL(k_cam_epi_fwd, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw, base_quat, frame_id, vid, M, V, out);
  return check_launch("camera_epilogue_forward");
}

extern "C" int lab4d_camera_epilogue_backward(const float* raw, const float* base_quat, const int64_t* frame_id, const int64_t* vid, const float* g_out, int M,
                                              int V, float* g_raw, float* row_scratch, float* g_base, int acc_base, void* stream) {
  EPI_CHECKS("camera_epilogue_backward");
  LAB4D_REQUIRE(raw && base_quat && g_out && g_raw && row_scratch, "camera_epilogue_backward: null pointer");
  hipLaunchKernelGGL(k_cam_epi_bwd, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw, base_quat, frame_id, vid, g_out, M, V, g_raw, row_scratch);
  if (g_base) hipLaunchKernelGGL(k_rows_to_video, dim3(V), dim3(256), 0, (hipStream_t)stream, (const float*)row_scratch, 4, frame_id, vid, M, V, g_base, acc_base);
  return check_launch("camera_epilogue_backward");
}

extern "C" int lab4d_intrinsics_epilogue_forward(const float* raw, const float* base_logfocal, const float* base_ppoint, const int64_t* frame_id, const int64_t* vid,
                                                 int M, int V, float* out, void* stream) {
  EPI_CHECKS("intrinsics_epilogue_forward");
  LAB4D_REQUIRE(raw && base_logfocal && base_ppoint && out, "intrinsics_epilogue_forward: null pointer");
  hipLaunchKernelGGL(k_intr_epi_fwd, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw, base_logfocal, base_ppoint, frame_id, vid, M, V, out);
  return check_launch("intrinsics_epilogue_forward");
}

extern "C" int lab4d_intrinsics_epilogue_backward(const float* raw, const float* base_logfocal, const int64_t* frame_id, const int64_t* vid, const float* g_out, int M,
                                                  int V, float* g_raw, float* row_scratch, float* g_video /* (V,4): [d logfocal | d ppoint] */, void* stream) {
  EPI_CHECKS("intrinsics_epilogue_backward");
  LAB4D_REQUIRE(raw && base_logfocal && g_out && g_raw && row_scratch, "intrinsics_epilogue_backward: null pointer");
  hipLaunchKernelGGL(k_intr_epi_bwd, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw, base_logfocal, frame_id, vid, g_out, M, V, g_raw, row_scratch);
  if (g_video) hipLaunchKernelGGL(k_rows_to_video, dim3(V), dim3(256), 0, (hipStream_t)stream, (const float*)row_scratch, 4, frame_id, vid, M, V, g_video, 0);
  return check_launch("intrinsics_epilogue_backward");
}
