// Optimizer step of the training loop on gfx950: global gradient norm + clip coefficient, and a
// multi-tensor AdamW over ONE flat fp32 buffer holding every parameter (one learning rate per parameter segment).
// Replaces torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW.step over ~200 one-parameter groups
// (lab4d/engine/trainer.py:164-190,581-604).  HBM-bound element-wise work: 16 B (gradient norm: 4 B) read and 12 B written per
// parameter; the point is the launch count -- two launches + one for the whole model instead of ~10 per group.
// Contract: include/lab4d_optim.h.
#include "common.hpp"
#include "optim_math.hpp"

namespace lab4d {
using namespace lab4d_optim;

constexpr int kNormBlocks = 512;

__global__ void __launch_bounds__(256) k_sumsq_partial(const float* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  const long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 x = g4[i];
    acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float x = g[(n4 << 2) + threadIdx.x];
    acc += x * x;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// norm = sqrt(sum partial); coef = min(1, max_norm / (norm + 1e-6))  (torch.nn.utils.clip_grad_norm_)
__global__ void __launch_bounds__(256) k_clip_coef(const float* __restrict__ partial, int nb, float max_norm, float* __restrict__ norm,
                                                    float* __restrict__ coef) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = sqrtf(red[0] + red[1] + red[2] + red[3]);
    norm[0] = t;
    coef[0] = fminf(1.f, max_norm / (t + 1e-6f));
  }
}

// check_grad (engine/trainer.py:581-604) on the device: the clip coefficient as above, and the reference's discard rule -- a step whose
// pre-clip norm exceeds skip_above is thrown away (the reference zeroes the gradients, which torch's optimizer then skips parameter by
// parameter: weights, moments and step counts stay untouched).  A non-finite norm is discarded as well (the reference would write NaN
// into every weight; its rollback then needs the host).  The device-side step counter advances only when the step is taken.
__global__ void __launch_bounds__(256) k_check_grad(const float* __restrict__ partial, int nb, float max_norm, float skip_above, float* __restrict__ norm,
                                                     float* __restrict__ coef, int* __restrict__ skipped, int* __restrict__ step) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = sqrtf(red[0] + red[1] + red[2] + red[3]);
    const bool skip = !(t <= skip_above);  // also true for NaN
    norm[0] = t;
    coef[0] = fminf(1.f, max_norm / (t + 1e-6f));
    skipped[0] = skip ? 1 : 0;
    if (!skip) step[0] += 1;
  }
}

// segments are padded to multiples of 4 elements, so a float4 never straddles two learning rates
__global__ void __launch_bounds__(256) k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                long n4, const long long* __restrict__ seg_end, const float* __restrict__ seg_lr, int nseg,
                                                AdamWHyper h, const float* __restrict__ grad_scale, const int* __restrict__ skipped,
                                                const int* __restrict__ dev_step, float beta1) {
  if (skipped && skipped[0]) return;  // discarded step: parameters and both moments untouched
  if (dev_step) {                     // bias corrections from the device-side count of the steps actually taken
    const int st = dev_step[0];
    h.bc1 = (float)(1.0 - pow((double)beta1, (double)st));
    h.bc2_sqrt = (float)sqrt(1.0 - pow((double)h.beta2, (double)st));
  }
  const float gs = grad_scale ? grad_scale[0] : 1.f;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float lr = seg_lr[segment_of(seg_end, nseg, i << 2)];
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    adamw_update(pp.x, gg.x * gs, mm.x, vv.x, lr, h);
    adamw_update(pp.y, gg.y * gs, mm.y, vv.y, lr, h);
    adamw_update(pp.z, gg.z * gs, mm.z, vv.z, lr, h);
    adamw_update(pp.w, gg.w * gs, mm.w, vv.w, lr, h);
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
}

}  // namespace lab4d

using namespace lab4d;

extern "C" int lab4d_grad_norm_clip(const float* g, int64_t n, float max_norm, float* work, float* norm, float* coef, void* stream) {
  LAB4D_REQUIRE(g && work && norm && coef, "grad_norm_clip: null pointer");
  LAB4D_REQUIRE(n >= 0 && max_norm > 0.f, "grad_norm_clip: bad n / max_norm");
  LAB4D_REQUIRE(((uintptr_t)g & 15) == 0, "grad_norm_clip: the gradient buffer must be 16-byte aligned");
  long blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > kNormBlocks) blocks = kNormBlocks;
  hipLaunchKernelGGL(k_sumsq_partial, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, g, (long)n, work);
  hipLaunchKernelGGL(k_clip_coef, dim3(1), dim3(256), 0, (hipStream_t)stream, work, (int)blocks, max_norm, norm, coef);
  return check_launch("grad_norm_clip");
}

extern "C" int lab4d_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end, const float* seg_lr, int nseg,
                                float beta1, float beta2, float eps, float weight_decay, int step, const float* grad_scale, void* stream) {
  LAB4D_REQUIRE(p && g && m && v && seg_end && seg_lr, "adamw_step: null pointer");
  LAB4D_REQUIRE(n >= 0 && n % 4 == 0 && nseg > 0 && step >= 1, "adamw_step: bad sizes (n=%ld must be a multiple of 4, nseg=%d, step=%d)", (long)n, nseg,
                step);
  LAB4D_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adamw_step: buffers must be 16-byte aligned");
  if (n == 0) return LAB4D_OK;
  AdamWHyper h;
  h.one_minus_beta1 = (float)(1.0 - (double)beta1);
  h.beta2 = beta2;
  h.one_minus_beta2 = (float)(1.0 - (double)beta2);
  h.eps = eps;
  h.weight_decay = weight_decay;
  h.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  h.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_adamw, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, (const long long*)seg_end, seg_lr, nseg, h,
                     grad_scale, (const int*)nullptr, (const int*)nullptr, beta1);
  return check_launch("adamw_step");
}

extern "C" int lab4d_check_grad(const float* g, int64_t n, float max_norm, float skip_above, float* work, float* norm, float* coef, int32_t* skipped,
                                int32_t* step, void* stream) {
  LAB4D_REQUIRE(g && work && norm && coef && skipped && step, "check_grad: null pointer");
  LAB4D_REQUIRE(n >= 0 && max_norm > 0.f && skip_above > 0.f, "check_grad: bad n / max_norm / skip_above");
  LAB4D_REQUIRE(((uintptr_t)g & 15) == 0, "check_grad: the gradient buffer must be 16-byte aligned");
  long blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > kNormBlocks) blocks = kNormBlocks;
  hipLaunchKernelGGL(k_sumsq_partial, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, g, (long)n, work);
  hipLaunchKernelGGL(k_check_grad, dim3(1), dim3(256), 0, (hipStream_t)stream, work, (int)blocks, max_norm, skip_above, norm, coef, skipped, step);
  return check_launch("check_grad");
}

extern "C" int lab4d_adamw_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end, const float* seg_lr, int nseg,
                                        float beta1, float beta2, float eps, float weight_decay, const float* grad_scale, const int32_t* skipped,
                                        const int32_t* step, void* stream) {
  LAB4D_REQUIRE(p && g && m && v && seg_end && seg_lr && skipped && step, "adamw_step_guarded: null pointer");
  LAB4D_REQUIRE(n >= 0 && n % 4 == 0 && nseg > 0, "adamw_step_guarded: bad sizes (n=%ld must be a multiple of 4, nseg=%d)", (long)n, nseg);
  LAB4D_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adamw_step_guarded: buffers must be 16-byte aligned");
  if (n == 0) return LAB4D_OK;
  AdamWHyper h;
  h.one_minus_beta1 = (float)(1.0 - (double)beta1);
  h.beta2 = beta2;
  h.one_minus_beta2 = (float)(1.0 - (double)beta2);
  h.eps = eps;
  h.weight_decay = weight_decay;
  h.bc1 = h.bc2_sqrt = 1.f;  // formed in the kernel from step[0]
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_adamw, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, (const long long*)seg_end, seg_lr, nseg, h,
                     grad_scale, (const int*)skipped, (const int*)step, beta1);
  return check_launch("adamw_step_guarded");
}
