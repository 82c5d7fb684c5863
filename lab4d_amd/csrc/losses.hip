// Per-ray loss epilogue (contract: include/lab4d_loss.h; reference: engine/model.py:401-611).  HBM-bound element-wise work over
// R rays (~40 floats read per ray each way): one grid-stride pass with wave + block reductions, 24 atomics per block.
#include "common.hpp"
#include "lab4d_loss.h"

namespace lab4d {

struct LossW {
  float w[LAB4D_LOSS_TERMS];
};

__device__ __forceinline__ float l2n(const float* a, const float* b, int n) {
  float s = 0.f;
  for (int i = 0; i < n; ++i) {
    const float d = a[i] - b[i];
    s += d * d;
  }
  return sqrtf(s);
}

// per-ray factors shared by forward and backward
struct RayMask {
  float m_mask, m_feat, m_dense, m_vd;  // m_vd = vis2d * detected (the comp opacity term carries no balance weight)
};
__device__ __forceinline__ RayMask ray_mask(const lab4d_loss_inputs& in, long r, int N) {
  const float det = in.t_detected ? in.t_detected[r / N] : 1.f;
  const float vis2d = in.t_vis2d ? in.t_vis2d[r] : 1.f;
  const float tm = in.t_mask ? in.t_mask[r] : 1.f;
  RayMask k;
  k.m_mask = (in.balance_wt ? in.balance_wt[r] : 1.f) * vis2d * det;
  k.m_vd = vis2d * det;
  k.m_feat = tm * det;
  k.m_dense = (in.dense_uses_mask ? tm : 1.f) * vis2d;
  return k;
}

__global__ void __launch_bounds__(256) k_ray_losses_fwd(lab4d_loss_inputs in, int R, int N, float* __restrict__ acc) {
  float s[LAB4D_LOSS_TERMS], c[LAB4D_LOSS_TERMS];
#pragma unroll
  for (int k = 0; k < LAB4D_LOSS_TERMS; ++k) s[k] = c[k] = 0.f;
  auto add = [&](int k, float v) {
    if (v > 0.f) { s[k] += v; c[k] += 1.f; }
  };
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (long)gridDim.x * blockDim.x) {
    const RayMask m = ray_mask(in, r, N);
    if (in.mask && in.t_mask) {
      const float d = in.mask[r] - in.t_mask[r];
      const float o = in.mask_all ? in.mask_all[r] - 1.f : 0.f;
      add(0, d * d * m.m_mask + o * o * m.m_vd);
    }
    if (in.feature) add(1, l2n(in.feature + r * 16, in.t_feature + r * 16, 16) * m.m_feat);
    if (in.xy_reproj) add(2, l2n(in.xy_reproj + r * 2, in.t_hxy + r * in.hxy_ld, 2) * m.m_feat);
    if (in.rgb)
      for (int j = 0; j < 3; ++j) { const float d = in.rgb[r * 3 + j] - in.t_rgb[r * 3 + j]; add(3, d * d * m.m_dense); }
    if (in.depth) add(4, fabsf(in.depth[r] - in.t_depth[r]) * m.m_dense);
    if (in.flow) add(5, l2n(in.flow + r * 2, in.t_flow + r * 2, 2) * (in.t_flow_uct[r] > 0.f ? 1.f : 0.f) * m.m_dense);
    if (in.vis) add(6, (in.vis[r] + (in.vis_bg ? in.vis_bg_wt * in.vis_bg[r] : 0.f)) * m.m_dense);
    if (in.gauss_mask && in.mask) { const float d = in.gauss_mask[r] - in.mask[r]; add(7, d * d); }
    if (in.eikonal) add(8, in.eikonal[r]);
    if (in.cyc_dist) add(9, in.cyc_dist[r]);
    if (in.delta_skin) add(10, in.delta_skin[r]);
    if (in.skin_entropy) add(11, in.skin_entropy[r]);
  }
  __shared__ float red[4][2 * LAB4D_LOSS_TERMS];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < LAB4D_LOSS_TERMS; ++k) {
    const float a = wave_sum(s[k]), b = wave_sum(c[k]);
    if (lane == 0) { red[wid][2 * k] = a; red[wid][2 * k + 1] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * LAB4D_LOSS_TERMS) {
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (v != 0.f) atomicAdd(acc + threadIdx.x, v);
  }
}

// loss[k] = w[k] * sum / count ; a term without positive elements is 0/0 = NaN in the reference (mean of an empty selection):
// kept, so that a degenerate batch shows up the same way
__global__ void k_ray_losses_finish(const float* __restrict__ acc, LossW w, unsigned present, float* __restrict__ loss) {
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int k = 0; k < LAB4D_LOSS_TERMS; ++k) {
      float v = 0.f;
      if ((present >> k) & 1u) v = w.w[k] * acc[2 * k] / acc[2 * k + 1];
      loss[k] = v;
      tot += v;
    }
    loss[LAB4D_LOSS_TERMS] = tot;
  }
}

__global__ void __launch_bounds__(256) k_ray_losses_bwd(lab4d_loss_inputs in, int R, int N, LossW w, const float* __restrict__ acc,
                                                         const float* __restrict__ g_loss, lab4d_loss_grads g) {
  __shared__ float coef[LAB4D_LOSS_TERMS];  // dL/d(sum of term k) = g_loss[k] * w[k] / count[k]
  if (threadIdx.x < LAB4D_LOSS_TERMS) coef[threadIdx.x] = g_loss[threadIdx.x] * w.w[threadIdx.x] / acc[2 * threadIdx.x + 1];
  __syncthreads();
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (long)gridDim.x * blockDim.x) {
    const RayMask m = ray_mask(in, r, N);
    float g_mask = 0.f, g_mask_all = 0.f;
    if (in.mask && in.t_mask) {
      const float d = in.mask[r] - in.t_mask[r];
      const float o = in.mask_all ? in.mask_all[r] - 1.f : 0.f;
      if (d * d * m.m_mask + o * o * m.m_vd > 0.f) {
        g_mask = 2.f * d * m.m_mask * coef[0];
        g_mask_all = 2.f * o * m.m_vd * coef[0];
      }
    }
    if (g.mask) g.mask[r] = g_mask;  // reg_gauss_mask sees the rendered mask detached (model.py:521)
    if (g.mask_all) g.mask_all[r] = g_mask_all;
    if (g.feature) {
      const float nrm = l2n(in.feature + r * 16, in.t_feature + r * 16, 16);
      const float f = (nrm * m.m_feat > 0.f) ? coef[1] * m.m_feat / nrm : 0.f;
      for (int j = 0; j < 16; ++j) g.feature[r * 16 + j] = f * (in.feature[r * 16 + j] - in.t_feature[r * 16 + j]);
    }
    if (g.xy_reproj) {
      const float nrm = l2n(in.xy_reproj + r * 2, in.t_hxy + r * in.hxy_ld, 2);
      const float f = (nrm * m.m_feat > 0.f) ? coef[2] * m.m_feat / nrm : 0.f;
      for (int j = 0; j < 2; ++j) g.xy_reproj[r * 2 + j] = f * (in.xy_reproj[r * 2 + j] - in.t_hxy[r * in.hxy_ld + j]);
    }
    if (g.rgb)
      for (int j = 0; j < 3; ++j) {
        const float d = in.rgb[r * 3 + j] - in.t_rgb[r * 3 + j];
        g.rgb[r * 3 + j] = (d * d * m.m_dense > 0.f) ? 2.f * d * m.m_dense * coef[3] : 0.f;
      }
    if (g.depth) {
      const float d = in.depth[r] - in.t_depth[r];
      g.depth[r] = (fabsf(d) * m.m_dense > 0.f) ? (d > 0.f ? 1.f : -1.f) * m.m_dense * coef[4] : 0.f;
    }
    if (g.flow) {
      const float nrm = l2n(in.flow + r * 2, in.t_flow + r * 2, 2);
      const float u = (in.t_flow_uct[r] > 0.f ? 1.f : 0.f) * m.m_dense;
      const float f = (nrm * u > 0.f) ? coef[5] * u / nrm : 0.f;
      for (int j = 0; j < 2; ++j) g.flow[r * 2 + j] = f * (in.flow[r * 2 + j] - in.t_flow[r * 2 + j]);
    }
    {
      const float v = (in.vis ? in.vis[r] : 0.f) + (in.vis_bg ? in.vis_bg_wt * in.vis_bg[r] : 0.f);
      const float gv = (in.vis && v * m.m_dense > 0.f) ? m.m_dense * coef[6] : 0.f;
      if (g.vis) g.vis[r] = gv;
      if (g.vis_bg) g.vis_bg[r] = gv * in.vis_bg_wt;
    }
    if (g.gauss_mask) {
      const float d = in.gauss_mask[r] - in.mask[r];
      g.gauss_mask[r] = (d * d > 0.f) ? 2.f * d * coef[7] : 0.f;
    }
    if (g.eikonal) g.eikonal[r] = in.eikonal[r] > 0.f ? coef[8] : 0.f;
    if (g.cyc_dist) g.cyc_dist[r] = in.cyc_dist[r] > 0.f ? coef[9] : 0.f;
    if (g.delta_skin) g.delta_skin[r] = in.delta_skin[r] > 0.f ? coef[10] : 0.f;
    if (g.skin_entropy) g.skin_entropy[r] = in.skin_entropy[r] > 0.f ? coef[11] : 0.f;
  }
}

static unsigned present_terms(const lab4d_loss_inputs& in) {
  const void* p[LAB4D_LOSS_TERMS] = {in.mask && in.t_mask ? in.mask : nullptr, in.feature, in.xy_reproj, in.rgb, in.depth, in.flow, in.vis,
                                     in.gauss_mask && in.mask ? in.gauss_mask : nullptr, in.eikonal, in.cyc_dist, in.delta_skin, in.skin_entropy};
  unsigned m = 0;
  for (int k = 0; k < LAB4D_LOSS_TERMS; ++k)
    if (p[k]) m |= 1u << k;
  return m;
}

static int check_inputs(const lab4d_loss_inputs* in, const char* what) {
  LAB4D_REQUIRE(in, "%s: null inputs", what);
  LAB4D_REQUIRE(!in->feature || in->t_feature, "%s: feature needs t_feature", what);
  LAB4D_REQUIRE(!in->xy_reproj || (in->t_hxy && in->hxy_ld >= 2), "%s: xy_reproj needs t_hxy", what);
  LAB4D_REQUIRE(!in->rgb || in->t_rgb, "%s: rgb needs t_rgb", what);
  LAB4D_REQUIRE(!in->depth || in->t_depth, "%s: depth needs t_depth", what);
  LAB4D_REQUIRE(!in->flow || (in->t_flow && in->t_flow_uct), "%s: flow needs t_flow and t_flow_uct", what);
  return LAB4D_OK;
}

}  // namespace lab4d
using namespace lab4d;

extern "C" int lab4d_ray_losses_forward(const lab4d_loss_inputs* in, int R, int N, const float* weights, float* acc, float* loss, void* stream) {
  if (int e = check_inputs(in, "ray_losses_forward")) return e;
  LAB4D_REQUIRE(R > 0 && N > 0 && weights && acc && loss, "ray_losses_forward: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (int e = zero_async(acc, 2 * LAB4D_LOSS_TERMS * sizeof(float), st)) return e;
  int grid = div_up(R, 256); if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(k_ray_losses_fwd, dim3(grid), dim3(256), 0, st, *in, R, N, acc);
  LossW w;
  for (int k = 0; k < LAB4D_LOSS_TERMS; ++k) w.w[k] = weights[k];
  hipLaunchKernelGGL(k_ray_losses_finish, dim3(1), dim3(64), 0, st, acc, w, present_terms(*in), loss);
  return check_launch("ray_losses_forward");
}

extern "C" int lab4d_ray_losses_backward(const lab4d_loss_inputs* in, int R, int N, const float* weights, const float* acc, const float* g_loss,
                                         const lab4d_loss_grads* g, void* stream) {
  if (int e = check_inputs(in, "ray_losses_backward")) return e;
  LAB4D_REQUIRE(R > 0 && N > 0 && weights && acc && g_loss && g, "ray_losses_backward: bad arguments");
  LossW w;
  for (int k = 0; k < LAB4D_LOSS_TERMS; ++k) w.w[k] = weights[k];
  int grid = div_up(R, 256); if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(k_ray_losses_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, *in, R, N, w, acc, g_loss, *g);
  return check_launch("ray_losses_backward");
}
