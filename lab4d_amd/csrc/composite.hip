// Alpha compositing along rays, forward + adjoint.
// Reference: lab4d/utils/render_utils.py:59-184 (render_pixel / compute_weights / integrate).
//
// One 64-lane wave per ray, lanes over samples (d = lane + 64 j): every per-sample field is read
// exactly once, coalesced; the transmittance prefix sum is a wave-level scan (shuffles) with a
// carry across 64-sample chunks; per-channel results are wave reductions.  HBM-bound:
// (2 + sum C) * 4 bytes read per sample, sum C * 4 bytes written per ray.
#include "common.hpp"

namespace lab4d {

constexpr int kMaxChunks = 4;  // D <= 256; kernels are instantiated for NC = ceil(D/64) in 1..4

template <int NC>
struct RayWeights {
  float tau[NC], w[NC], T[NC];  // per-lane samples
  float mask;                   // sum_d w
};

// tau = sigma*delta ; T_d = exp(-cumsum tau) ; w_d = (1 - exp(-tau_d)) * T_{d-1}   (render_utils.py:99-126)
template <int NC>
__device__ __forceinline__ void ray_weights(const float* density, const float* deltas, long base, int D, int lane,
                                            RayWeights<NC>& rw) {
  float carry = 0.f;
  float msum = 0.f;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int d = lane + 64 * j;
    float tau = 0.f;
    if (d < D) tau = density[base + d] * deltas[base + d];
    const float incl = wave_scan_incl(tau, lane) + carry;
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = carry;
    carry = __shfl(incl, 63, 64);
    const float T = expf(-incl);
    const float Tex = expf(-excl);
    const float w = (d < D) ? (1.f - expf(-tau)) * Tex : 0.f;
    rw.tau[j] = tau; rw.w[j] = w; rw.T[j] = T;
    msum += w;
  }
  rw.mask = wave_sum(msum);
}


// ---- per-sample fields with C channels, read / written as ONE contiguous (D*C)-float run per ray ----------------------------
// The first version walked a field channel by channel (lane d reads v[(base+d)*C + c]: a 4*C-byte stride, every line touched C times
// and every gradient line written C times in 4-byte pieces): k_composite_bwd ran at 1.0 TB/s.  Here lane l handles the flattened
// elements e = l, l+64, ... (d = e / C, c = e % C), the per-sample weights come from a wave-private LDS copy, and the per-sample
// reduction over the channels (dL/dw_hat) goes through LDS (shuffles when C divides 64, ds_add_f32 otherwise).
constexpr int kMaxD = 64 * 4;
// e / C for e < 2^15 without an integer division: shift for powers of two, multiply-shift for 3 (exact below 2^16)
__device__ __forceinline__ int div_c(int e, int C, int sh) { return sh >= 0 ? (e >> sh) : (C == 3 ? (int)(((unsigned)e * 43691u) >> 17) : e / C); }
__device__ __forceinline__ int shift_of(int C) { return (C & (C - 1)) == 0 ? __ffs(C) - 1 : -1; }

// out_c = sum_d wl[d] * v[d][c]  -> written by the lanes c < C
template <int NC>
__device__ __forceinline__ void field_reduce(const float* __restrict__ v, int D, int C, const float* wl, int lane, float* __restrict__ out_c) {
  const int n = D * C, sh = shift_of(C);
  if ((64 % C) == 0) {
    float s = 0.f;
    for (int e = lane; e < n; e += 64) s += wl[e >> sh] * v[e];
    for (int o = C; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
    if (lane < C) out_c[lane] = s;
  } else if (C == 3) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int e = lane; e < n; e += 64) {
      const int d = div_c(e, 3, -1), c = e - 3 * d;
      const float x = wl[d] * v[e];
      s0 += c == 0 ? x : 0.f; s1 += c == 1 ? x : 0.f; s2 += c == 2 ? x : 0.f;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { out_c[0] = s0; out_c[1] = s1; out_c[2] = s2; }
  } else {
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
      for (int d = lane; d < D; d += 64) s += wl[d] * v[d * C + c];
      s = wave_sum(s);
      if (lane == 0) out_c[c] = s;
    }
  }
}

template <int NC>
__global__ void __launch_bounds__(256) k_composite_fwd(const float* __restrict__ density, const float* __restrict__ deltas,
                                                        lab4d_field_list fl, const float* __restrict__ flow,
                                                        const float* __restrict__ vis, const float* __restrict__ gdens, int R, int D,
                                                        int sumC, float* __restrict__ weights, float* __restrict__ transmit,
                                                        float* __restrict__ mask, float* __restrict__ out,
                                                        float* __restrict__ flow_out, float* __restrict__ vis_num,
                                                        float* __restrict__ t_sum, float* __restrict__ gauss_mask) {
  const int lane = threadIdx.x & 63;
  constexpr int nchunk = NC;
  __shared__ float wl_all[4][64 * NC];
  float* wl = wl_all[threadIdx.x >> 6];  // wave-private: normalised weights of the current ray
  for (long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6); ray < R; ray += (long)gridDim.x * 4) {
    const long base = ray * D;
    RayWeights<NC> rw;
    ray_weights(density, deltas, base, D, lane, rw);
    const float inv = 1.f / (rw.mask + 1e-6f);
    _Pragma("unroll") for (int j = 0; j < nchunk; ++j) wl[lane + 64 * j] = rw.w[j] * inv;
    __builtin_amdgcn_wave_barrier();
    if (weights || transmit) {
      _Pragma("unroll") for (int j = 0; j < nchunk; ++j) {
        const int d = lane + 64 * j;
        if (d < D) {
          if (weights) weights[base + d] = rw.w[j];
          if (transmit) transmit[base + d] = rw.T[j];
        }
      }
    }
    if (lane == 0 && mask) mask[ray] = rw.mask;
    int co = 0;
    for (int f = 0; f < fl.n_fields; ++f) {
      const int C = fl.channels[f];
      const float* v = fl.fields[f];
      const int mode = fl.modes[f];
      const float* vr = v + base * C;  // this ray's (D, C) block: D*C contiguous floats
      if (mode == 2) {  // mean over (D, C)
        float s = 0.f;
        for (int e = lane; e < D * C; e += 64) s += vr[e];
        s = wave_sum(s);
        if (lane == 0) out[ray * sumC + co] = s / (float)(D * C);
        co += 1;
      } else {
        field_reduce<NC>(vr, D, C, wl, lane, out + ray * sumC + co);
        co += C;
      }
    }
    if (flow) {  // render_utils.py:160-167: weights gated by the validity channel, re-normalised
      float sw = 0.f, su = 0.f, sv = 0.f;
      _Pragma("unroll") for (int j = 0; j < nchunk; ++j) {
        const int d = lane + 64 * j;
        if (d < D) {
          const float* f3 = flow + (base + d) * 3;
          const float wf = rw.w[j] * f3[2];
          sw += wf; su += wf * f3[0]; sv += wf * f3[1];
        }
      }
      sw = wave_sum(sw); su = wave_sum(su); sv = wave_sum(sv);
      if (lane == 0) {
        const float i2 = 1.f / (sw + 1e-6f);
        flow_out[ray * 2] = su * i2; flow_out[ray * 2 + 1] = sv * i2;
      }
    }
    if (vis) {  // -(logsigmoid(vis) * T).mean(D)  (render_utils.py:82-90); the /mean(T) is done by the caller
      float s = 0.f, ts = 0.f;
      _Pragma("unroll") for (int j = 0; j < nchunk; ++j) {
        const int d = lane + 64 * j;
        if (d < D) {
          const float x = vis[base + d];
          const float ls = fminf(x, 0.f) - log1pf(expf(-fabsf(x)));  // logsigmoid
          s += ls * rw.T[j]; ts += rw.T[j];
        }
      }
      s = wave_sum(s); ts = wave_sum(ts);
      if (lane == 0) { vis_num[ray] = -s / (float)D; t_sum[ray] = ts; }
    }
    if (gdens) {  // render_utils.py:93-95: mask of the gaussian-bone density
      RayWeights<NC> g;
      ray_weights(gdens, deltas, base, D, lane, g);
      if (lane == 0) gauss_mask[ray] = g.mask;
    }
  }
}

template <int NC>
__global__ void __launch_bounds__(256) k_composite_bwd(const float* __restrict__ density, const float* __restrict__ deltas,
                                                        lab4d_field_list fl, const float* __restrict__ flow,
                                                        const float* __restrict__ vis, const float* __restrict__ gdens, int R, int D,
                                                        int sumC, const float* __restrict__ g_mask, const float* __restrict__ g_out,
                                                        const float* __restrict__ g_flow_out, const float* __restrict__ g_vis_num,
                                                        const float* __restrict__ g_gauss_mask, float* __restrict__ g_density,
                                                        float* __restrict__ g_deltas, lab4d_field_grads gf,
                                                        float* __restrict__ g_flow, float* __restrict__ g_vis,
                                                        float* __restrict__ g_gdens) {
  const int lane = threadIdx.x & 63;
  constexpr int nchunk = NC;
  __shared__ float wl_all[4][64 * NC], al_all[4][64 * NC];
  float* wl = wl_all[threadIdx.x >> 6];  // wave-private: normalised weights w_d / Z of the current ray
  float* al = al_all[threadIdx.x >> 6];  // wave-private: dL/d(w_hat)_d accumulated over the mode-0 channels
  for (long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6); ray < R; ray += (long)gridDim.x * 4) {
    const long base = ray * D;
    RayWeights<NC> rw;
    ray_weights(density, deltas, base, D, lane, rw);
    const float Z = rw.mask + 1e-6f, inv = 1.f / Z;
    float gw[NC];   // dL/dw_d
    float A[NC];    // dL/d(w_hat)_d = sum over mode-0 channels of g_c v_dc
#pragma unroll
    for (int j = 0; j < NC; ++j) { gw[j] = 0.f; A[j] = 0.f; wl[lane + 64 * j] = rw.w[j] * inv; al[lane + 64 * j] = 0.f; }
    __builtin_amdgcn_wave_barrier();
    int co = 0;
    for (int f = 0; f < fl.n_fields; ++f) {
      const int C = fl.channels[f];
      const float* v = fl.fields[f];
      float* gv = gf.fields[f];
      const int mode = fl.modes[f];
      const float* vr = v + base * C;
      float* gvr = gv ? gv + base * C : nullptr;
      const int n = D * C;
      if (mode == 2) {
        const float g = g_out ? g_out[ray * sumC + co] / (float)(D * C) : 0.f;
        if (gvr) for (int e = lane; e < n; e += 64) gvr[e] = g;
        co += 1;
      } else {
        const float* gc = g_out ? g_out + ray * sumC + co : nullptr;
        const int sh = shift_of(C);
        if ((64 % C) == 0) {
          const int c = lane & (C - 1);
          const float g = gc ? gc[c] : 0.f;
          for (int e = lane; e < n; e += 64) {
            const int d = e >> sh;
            if (gvr) gvr[e] = g * wl[d];
            if (mode == 0) {
              float a = g * vr[e];
              for (int o = 1; o < C; o <<= 1) a += __shfl_xor(a, o, 64);  // the C lanes of one sample
              if (c == 0) al[d] += a;
            }
          }
        } else {
          for (int e = lane; e < n; e += 64) {
            const int d = div_c(e, C, sh), c = e - d * C;
            const float g = gc ? gc[c] : 0.f;
            if (gvr) gvr[e] = g * wl[d];
            if (mode == 0) atomicAdd(&al[d], g * vr[e]);  // ds_add_f32
          }
        }
        co += C;
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NC; ++j) A[j] = al[lane + 64 * j];
    // w_hat = w / Z:  dL/dw_d = A_d / Z - (sum_j A_j w_j) / Z^2 + g_mask
    float aw = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j) aw += A[j] * rw.w[j];
    aw = wave_sum(aw);
    const float gm = g_mask ? g_mask[ray] : 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j) gw[j] = A[j] * inv - aw * inv * inv + gm;
    if (flow) {
      float sw = 0.f, sb = 0.f;
      float Bv[NC], val[NC];
      const float gu = g_flow_out ? g_flow_out[ray * 2] : 0.f, gvv = g_flow_out ? g_flow_out[ray * 2 + 1] : 0.f;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const int d = lane + 64 * j;
        Bv[j] = 0.f; val[j] = 0.f;
        if (j < nchunk && d < D) {
          const float* f3 = flow + (base + d) * 3;
          val[j] = f3[2];
          Bv[j] = gu * f3[0] + gvv * f3[1];
          const float wf = rw.w[j] * f3[2];
          sw += wf; sb += Bv[j] * wf;
        }
      }
      sw = wave_sum(sw); sb = wave_sum(sb);
      const float F = sw + 1e-6f, iF = 1.f / F;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const int d = lane + 64 * j;
        if (j < nchunk && d < D) {
          gw[j] += val[j] * (Bv[j] * iF - sb * iF * iF);
          if (g_flow) {
            const float wn = rw.w[j] * val[j] * iF;
            g_flow[(base + d) * 3] = wn * gu; g_flow[(base + d) * 3 + 1] = wn * gvv; g_flow[(base + d) * 3 + 2] = 0.f;
          }
        }
      }
    }
    // w_d = (1-exp(-tau_d)) Texcl_d :  dL/dtau_d = gw_d * T_d  -  sum_{i>d} gw_i w_i   (suffix scan)
    float carry = 0.f;
    float gtau[NC];
    _Pragma("unroll") for (int j = nchunk - 1; j >= 0; --j) {
      const float x = gw[j] * rw.w[j];
      const float incl = wave_rscan_incl(x, lane) + carry;
      const float excl = incl - x;
      carry = __shfl(incl, 0, 64);
      gtau[j] = gw[j] * rw.T[j] - excl;
    }
    float gdl[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) gdl[j] = 0.f;
    if (gdens && g_gauss_mask) {  // sum_d w^g_d = 1 - T^g_last  ->  d/dtau^g_d = T^g_last
      RayWeights<NC> g;
      ray_weights(gdens, deltas, base, D, lane, g);
      float last = 0.f;
      const int jl = (D - 1) >> 6, ll = (D - 1) & 63;
#pragma unroll
      for (int j = 0; j < NC; ++j) if (j == jl) last = __shfl(g.T[j], ll, 64);
      const float gt = g_gauss_mask[ray] * last;
      _Pragma("unroll") for (int j = 0; j < nchunk; ++j) {
        const int d = lane + 64 * j;
        if (d < D) {
          if (g_gdens) g_gdens[base + d] = gt * deltas[base + d];
          gdl[j] = gt * gdens[base + d];
        }
      }
    } else if (g_gdens) {
      _Pragma("unroll") for (int j = 0; j < nchunk; ++j) { const int d = lane + 64 * j; if (d < D) g_gdens[base + d] = 0.f; }
    }
    _Pragma("unroll") for (int j = 0; j < nchunk; ++j) {
      const int d = lane + 64 * j;
      if (d < D) {
        if (g_density) g_density[base + d] = gtau[j] * deltas[base + d];
        if (g_deltas) g_deltas[base + d] = gtau[j] * density[base + d] + gdl[j];
      }
    }
    if (vis && g_vis) {
      const float g = g_vis_num ? g_vis_num[ray] : 0.f;
      _Pragma("unroll") for (int j = 0; j < nchunk; ++j) {
        const int d = lane + 64 * j;
        if (d < D) {
          const float x = vis[base + d];
          const float sneg = 1.f / (1.f + expf(x));  // sigmoid(-x) = d logsigmoid / dx
          g_vis[base + d] = -g * rw.T[j] * sneg / (float)D;
        }
      }
    }
  }
}

}  // namespace lab4d
using namespace lab4d;

static int check_fields(const lab4d_field_list* fl, int* sumC) {
  LAB4D_REQUIRE(fl && fl->n_fields >= 0 && fl->n_fields <= LAB4D_MAX_FIELDS, "composite: bad field list");
  int s = 0;
  for (int i = 0; i < fl->n_fields; ++i) {
    LAB4D_REQUIRE(fl->fields[i] && fl->channels[i] >= 1 && fl->modes[i] >= 0 && fl->modes[i] <= 2, "composite: bad field %d", i);
    s += fl->modes[i] == 2 ? 1 : fl->channels[i];
  }
  *sumC = s;
  return LAB4D_OK;
}

extern "C" int lab4d_composite_forward(const float* density, const float* deltas, const lab4d_field_list* fl, const float* flow,
                                       const float* vis, const float* gauss_density, int R, int D, float* weights,
                                       float* transmit, float* mask, float* out, float* flow_out, float* vis_num,
                                       float* t_sum, float* gauss_mask, void* stream) {
  LAB4D_REQUIRE(density && deltas, "composite_forward: null density/deltas");
  LAB4D_REQUIRE(D >= 1 && D <= 64 * kMaxChunks, "composite_forward: D must be in [1,%d] (got %d)", 64 * kMaxChunks, D);
  int sumC = 0;
  if (int e = check_fields(fl, &sumC)) return e;
  LAB4D_REQUIRE(sumC == 0 || out, "composite_forward: out is null");
  LAB4D_REQUIRE(!flow || flow_out, "composite_forward: flow_out is null");
  LAB4D_REQUIRE(!vis || (vis_num && t_sum), "composite_forward: vis_num/t_sum is null");
  LAB4D_REQUIRE(!gauss_density || gauss_mask, "composite_forward: gauss_mask is null");
  if (R == 0) return LAB4D_OK;
  int grid = div_up(R, 4); if (grid > 16384) grid = 16384;
#define LAUNCH_FWD(NC) hipLaunchKernelGGL((k_composite_fwd<NC>), dim3(grid), dim3(256), 0, (hipStream_t)stream, density, deltas, *fl, \
                     flow, vis, gauss_density, R, D, sumC, weights, transmit, mask, out, flow_out, vis_num, t_sum, gauss_mask)
  switch ((D + 63) / 64) { case 1: LAUNCH_FWD(1); break; case 2: LAUNCH_FWD(2); break; case 3: LAUNCH_FWD(3); break; default: LAUNCH_FWD(4); }
  return check_launch("composite_forward");
}

extern "C" int lab4d_composite_backward(const float* density, const float* deltas, const lab4d_field_list* fl, const float* flow,
                                        const float* vis, const float* gauss_density, int R, int D, const float* g_mask,
                                        const float* g_out, const float* g_flow_out, const float* g_vis_num,
                                        const float* g_gauss_mask, float* g_density, float* g_deltas,
                                        const lab4d_field_grads* g_fields, float* g_flow, float* g_vis, float* g_gauss_density,
                                        void* stream) {
  LAB4D_REQUIRE(density && deltas, "composite_backward: null density/deltas");
  LAB4D_REQUIRE(D >= 1 && D <= 64 * kMaxChunks, "composite_backward: D must be in [1,%d]", 64 * kMaxChunks);
  int sumC = 0;
  if (int e = check_fields(fl, &sumC)) return e;
  LAB4D_REQUIRE(g_fields && g_fields->n_fields == fl->n_fields, "composite_backward: g_fields does not match the field list");
  if (R == 0) return LAB4D_OK;
  int grid = div_up(R, 4); if (grid > 16384) grid = 16384;
#define LAUNCH_BWD(NC) hipLaunchKernelGGL((k_composite_bwd<NC>), dim3(grid), dim3(256), 0, (hipStream_t)stream, density, deltas, *fl, \
                     flow, vis, gauss_density, R, D, sumC, g_mask, g_out, g_flow_out, g_vis_num, g_gauss_mask, g_density, \
                     g_deltas, *g_fields, g_flow, g_vis, g_gauss_density)
  switch ((D + 63) / 64) { case 1: LAUNCH_BWD(1); break; case 2: LAUNCH_BWD(2); break; case 3: LAUNCH_BWD(3); break; default: LAUNCH_BWD(4); }
  return check_launch("composite_backward");
}
