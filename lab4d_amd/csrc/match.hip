// FeatureNeRF.global_match (lab4d/nnutils/feature.py:152-199): soft arg-max of pixel features over K <= 1024 canonical candidates,
//   score = (feat_px @ feat_c^T) * exp(logsigma);  prob = softmax(score, 1);  xyz_matched = prob @ xyz_c
// as one kernel forward and two backward (the reference's matmul / mul / softmax / matmul and their adjoints stream the (R, K) score
// matrix eight times: 4 GB per 131,072 rays; here it never exists).  Contract: include/lab4d_hip.h (section 3d).
// Forward / backward-by-ray: one thread per ray, the candidates (K x 16 features, K x 3 points) in LDS, read as broadcasts.
// Backward-by-candidate: a resident grid; each thread owns K/256 candidates in registers and walks its block's rays (staged through LDS);
// per-block partials go to a workspace and are summed by a last small kernel (no atomics: 19 K floats per block).
#include "common.hpp"

namespace lab4d {

constexpr int MC = 16;       // feature channels
constexpr int MKMAX = 1024;  // candidates

__device__ __forceinline__ float dot16(const float (&f)[MC], const float4* row) {
  const float4 a = row[0], b = row[1], c = row[2], d = row[3];
  float s = f[0] * a.x;
  s += f[1] * a.y; s += f[2] * a.z; s += f[3] * a.w;
  s += f[4] * b.x; s += f[5] * b.y; s += f[6] * b.z; s += f[7] * b.w;
  s += f[8] * c.x; s += f[9] * c.y; s += f[10] * c.z; s += f[11] * c.w;
  s += f[12] * d.x; s += f[13] * d.y; s += f[14] * d.z; s += f[15] * d.w;
  return s;
}

// score = (f . fc) * scale as a ROUNDED product: contracted into fma(dot, scale, -max) the subtraction would see the unrounded product in one
// pass and the rounded one in the other (the row maximum would not be the maximum of the scores it is subtracted from)
// (__fmul_rn alone is still fused by the backend under -ffp-contract=fast, see hashgrid_math.hpp: the product is made opaque instead)
__device__ __forceinline__ float score(float d, float scale) {
  float p = d * scale;
  asm volatile("" : "+v"(p));
  return p;
}

// out[r] = sum_i softmax_i(scale * f_r . fc_i) xc_i ; stats[r] = (max_i score, sum_i exp(score - max)).  One thread per ray; the candidates pass
// through LDS in tiles of KT, twice: the row maximum first (the reference's softmax subtracts it before exponentiating), then the sums.
constexpr int KT = 512;
__global__ void __launch_bounds__(256) k_match_fwd(const float* __restrict__ feat_px, const float* __restrict__ fc, const float* __restrict__ xc,
                                                   const float* __restrict__ logsigma, int R, int K, float* __restrict__ out,
                                                   float* __restrict__ stats) {
  __shared__ float4 sfc[KT * 4];
  __shared__ float4 sxc[KT];
  const float scale = expf(*logsigma);
  const int r = blockIdx.x * 256 + threadIdx.x;
  const int rc = r < R ? r : R - 1;
  float f[MC];
#pragma unroll
  for (int c = 0; c < MC; c += 4) {
    const float4 v = *reinterpret_cast<const float4*>(feat_px + (size_t)rc * MC + c);
    f[c] = v.x; f[c + 1] = v.y; f[c + 2] = v.z; f[c + 3] = v.w;
  }
  float m = -INFINITY, l = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int pass = 0; pass < 2; ++pass)
    for (int k0 = 0; k0 < K; k0 += KT) {
      const int nk = min(KT, K - k0);
      __syncthreads();
      for (int e = threadIdx.x; e < nk * 4; e += 256) sfc[e] = reinterpret_cast<const float4*>(fc + (size_t)k0 * MC)[e];
      if (pass == 1)
        for (int e = threadIdx.x; e < nk; e += 256) sxc[e] = make_float4(xc[3 * (size_t)(k0 + e)], xc[3 * (size_t)(k0 + e) + 1], xc[3 * (size_t)(k0 + e) + 2], 0.f);
      __syncthreads();
      if (pass == 0) {
        for (int i = 0; i < nk; ++i) m = fmaxf(m, score(dot16(f, sfc + 4 * i), scale));
      } else {
        for (int i = 0; i < nk; ++i) {
          const float e = expf(score(dot16(f, sfc + 4 * i), scale) - m);
          const float4 x = sxc[i];
          l += e; a0 += e * x.x; a1 += e * x.y; a2 += e * x.z;
        }
      }
    }
  if (r < R) {
    const float inv = 1.f / l;
    out[(size_t)r * 3] = a0 * inv; out[(size_t)r * 3 + 1] = a1 * inv; out[(size_t)r * 3 + 2] = a2 * inv;
    stats[(size_t)r * 2] = m; stats[(size_t)r * 2 + 1] = l;
  }
}

// Adjoint, candidate side.  With p_ri the softmax:  ds_ri = p_ri g_r . (xc_i - out_r)   (= p_ri (g_r . xc_i - sum_j p_rj g_r . xc_j), formed as
// one difference so that it vanishes exactly where the softmax is constant);
//   g_fc_i = scale sum_r ds_ri f_r ;  g_xc_i = sum_r p_ri g_r ;  g_logsigma = sum_ri ds_ri score_ri.
// Thread t of a block owns candidates t, t + 256, ... (NCT = K / 256 of them); the block's rays pass through LDS 64 at a time.
template <int NCT>
__global__ void __launch_bounds__(256) k_match_bwd_cand(const float* __restrict__ feat_px, const float* __restrict__ fc, const float* __restrict__ xc,
                                                        const float* __restrict__ logsigma, const float* __restrict__ out,
                                                        const float* __restrict__ stats, const float* __restrict__ g_out, int R, int K,
                                                        float* __restrict__ part /* (gridDim.x, K, 20) */) {
  constexpr int TR = 64, RW = 24;  // rays per LDS tile, floats per staged ray: f[16] g[3] m 1/l out[3]
  __shared__ float ray[TR * RW];
  __shared__ float red[256];
  const float scale = expf(*logsigma);
  float cf[NCT][MC], cx[NCT][3], gf[NCT][MC], gx[NCT][3];
  float gls = 0.f;
#pragma unroll
  for (int j = 0; j < NCT; ++j) {
    const int i = threadIdx.x + 256 * j;
#pragma unroll
    for (int c = 0; c < MC; ++c) { cf[j][c] = i < K ? fc[(size_t)i * MC + c] : 0.f; gf[j][c] = 0.f; }
#pragma unroll
    for (int c = 0; c < 3; ++c) { cx[j][c] = i < K ? xc[(size_t)i * 3 + c] : 0.f; gx[j][c] = 0.f; }
  }
  const int per = (R + gridDim.x - 1) / gridDim.x;
  const int r_begin = blockIdx.x * per, r_end = min(R, r_begin + per);
  for (int r0 = r_begin; r0 < r_end; r0 += TR) {
    const int nr = min(TR, r_end - r0);
    __syncthreads();
    for (int e = threadIdx.x; e < nr * MC; e += 256) ray[(e / MC) * RW + (e % MC)] = feat_px[(size_t)r0 * MC + e];
    if (threadIdx.x < nr) {
      const int r = r0 + threadIdx.x;
      const float g0 = g_out[(size_t)r * 3], g1 = g_out[(size_t)r * 3 + 1], g2 = g_out[(size_t)r * 3 + 2];
      float* q = ray + threadIdx.x * RW + MC;
      q[0] = g0; q[1] = g1; q[2] = g2;
      q[3] = stats[(size_t)r * 2];
      q[4] = 1.f / stats[(size_t)r * 2 + 1];
      q[5] = out[(size_t)r * 3]; q[6] = out[(size_t)r * 3 + 1]; q[7] = out[(size_t)r * 3 + 2];
    }
    __syncthreads();
    for (int k = 0; k < nr; ++k) {
      const float* q = ray + k * RW;
      float f[MC];
#pragma unroll
      for (int c = 0; c < MC; ++c) f[c] = q[c];
      const float g0 = q[MC], g1 = q[MC + 1], g2 = q[MC + 2], m = q[MC + 3], invl = q[MC + 4], o0 = q[MC + 5], o1 = q[MC + 6], o2 = q[MC + 7];
#pragma unroll
      for (int j = 0; j < NCT; ++j) {
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < MC; ++c) d += f[c] * cf[j][c];
        const float s = score(d, scale);
        const float p = expf(s - m) * invl;
        const float ds = p * (g0 * (cx[j][0] - o0) + g1 * (cx[j][1] - o1) + g2 * (cx[j][2] - o2));
#pragma unroll
        for (int c = 0; c < MC; ++c) gf[j][c] += ds * f[c];
        gx[j][0] += p * g0; gx[j][1] += p * g1; gx[j][2] += p * g2;
        gls += ds * s;
      }
    }
  }
  // block partials: (K, 20) = [g_fc (16, scaled) | g_xc (3) | g_logsigma share]
  red[threadIdx.x] = gls;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NCT; ++j) {
    const int i = threadIdx.x + 256 * j;
    if (i < K) {
      float* o = part + ((size_t)blockIdx.x * K + i) * 20;
#pragma unroll
      for (int c = 0; c < MC; ++c) o[c] = gf[j][c] * scale;
      o[16] = gx[j][0]; o[17] = gx[j][1]; o[18] = gx[j][2];
      o[19] = i == 0 ? red[0] : 0.f;
    }
  }
}

// g_fc (K,16), g_xc (K,3), g_logsigma (1) = sums of the per-block partials (fixed order: deterministic)
__global__ void __launch_bounds__(256) k_match_bwd_reduce(const float* __restrict__ part, int NB, int K, float* __restrict__ g_fc,
                                                          float* __restrict__ g_xc, float* __restrict__ g_ls) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= K * 20) return;
  float s = 0.f;
  for (int b = 0; b < NB; ++b) s += part[(size_t)b * K * 20 + e];
  const int i = e / 20, c = e - 20 * i;
  if (c < MC) g_fc[(size_t)i * MC + c] = s;
  else if (c < 19) g_xc[(size_t)i * 3 + (c - MC)] = s;
  else if (i == 0) *g_ls = s;
}

}  // namespace lab4d
using namespace lab4d;

static int match_blocks() { return 256; }

extern "C" int lab4d_global_match_workspace_floats(int K) { return match_blocks() * K * 20; }

extern "C" int lab4d_global_match_forward(const float* feat_px, const float* feat_c, const float* xyz_c, const float* logsigma, int R, int C,
                                          int K, float* out, float* stats, void* stream) {
  LAB4D_REQUIRE(feat_px && feat_c && xyz_c && logsigma && out && stats, "global_match_forward: null pointer");
  LAB4D_REQUIRE(C == MC && K >= 1 && K <= MKMAX && R >= 0, "global_match_forward: C must be 16 and 1 <= K <= 1024 (C=%d K=%d)", C, K);
  LAB4D_REQUIRE((((uintptr_t)feat_px | (uintptr_t)feat_c) & 15) == 0, "global_match_forward: feat_px / feat_c must be 16-byte aligned (rows are read as float4)");
  if (R == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_match_fwd, dim3(div_up(R, 256)), dim3(256), 0, (hipStream_t)stream, feat_px, feat_c, xyz_c, logsigma, R, K, out, stats);
  return check_launch("global_match_forward");
}

extern "C" int lab4d_global_match_backward(const float* feat_px, const float* feat_c, const float* xyz_c, const float* logsigma,
                                           const float* out, const float* stats, const float* g_out, int R, int C, int K, float* g_feat_c,
                                           float* g_xyz_c, float* g_logsigma, float* work, void* stream) {
  LAB4D_REQUIRE(feat_px && feat_c && xyz_c && logsigma && out && stats && g_out && g_feat_c && g_xyz_c && g_logsigma && work,
                "global_match_backward: null pointer");
  LAB4D_REQUIRE(C == MC && K >= 1 && K <= MKMAX && R >= 0, "global_match_backward: C must be 16 and 1 <= K <= 1024 (C=%d K=%d)", C, K);
  LAB4D_REQUIRE((((uintptr_t)feat_px | (uintptr_t)feat_c) & 15) == 0, "global_match_backward: feat_px / feat_c must be 16-byte aligned (rows are read as float4)");
  hipStream_t st = (hipStream_t)stream;
  int NB = match_blocks();
  if (R == 0) NB = 0;
  if (NB > 0) {
    if (NB > div_up(R, 64)) NB = div_up(R, 64);
    const int nct = div_up(K, 256);
    if (nct == 1) hipLaunchKernelGGL((k_match_bwd_cand<1>), dim3(NB), dim3(256), 0, st, feat_px, feat_c, xyz_c, logsigma, out, stats, g_out, R, K, work);
    else if (nct == 2) hipLaunchKernelGGL((k_match_bwd_cand<2>), dim3(NB), dim3(256), 0, st, feat_px, feat_c, xyz_c, logsigma, out, stats, g_out, R, K, work);
    else if (nct == 3) hipLaunchKernelGGL((k_match_bwd_cand<3>), dim3(NB), dim3(256), 0, st, feat_px, feat_c, xyz_c, logsigma, out, stats, g_out, R, K, work);
    else hipLaunchKernelGGL((k_match_bwd_cand<4>), dim3(NB), dim3(256), 0, st, feat_px, feat_c, xyz_c, logsigma, out, stats, g_out, R, K, work);
    if (int e = check_launch("global_match_backward")) return e;
  }
  hipLaunchKernelGGL(k_match_bwd_reduce, dim3(div_up(K * 20, 256)), dim3(256), 0, st, work, NB, K, g_feat_c, g_xyz_c, g_logsigma);
  return check_launch("global_match_backward(reduce)");
}
