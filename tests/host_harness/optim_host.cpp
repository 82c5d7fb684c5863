// CPU build of lab4d_amd/csrc/optim_math.hpp for the test-suite (see fk_host.cpp): the per-element AdamW arithmetic and the
// segment lookup of csrc/optim.hip.  TEST INFRASTRUCTURE ONLY.
#include "optim_math.hpp"

using namespace lab4d_optim;

extern "C" int adamw_host_step(float* p, const float* g, float* m, float* v, long long n, const long long* seg_end, const float* seg_lr, int nseg,
                               float beta1, float beta2, float eps, float wd, int step, float grad_scale) {
  AdamWHyper h;
  h.one_minus_beta1 = (float)(1.0 - (double)beta1);
  h.beta2 = beta2;
  h.one_minus_beta2 = (float)(1.0 - (double)beta2);
  h.eps = eps;
  h.weight_decay = wd;
  h.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  h.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  for (long long e = 0; e < n; ++e) adamw_update(p[e], g[e] * grad_scale, m[e], v[e], seg_lr[segment_of(seg_end, nseg, e & ~3LL)], h);
  return 0;
}

// same contract as lab4d_grad_norm_clip (a plain loop; the reduction tree of the kernel is exercised on the GPU only)
extern "C" int grad_norm_clip_host(const float* g, long long n, float max_norm, float* norm, float* coef) {
  double acc = 0.0;
  for (long long e = 0; e < n; ++e) acc += (double)g[e] * g[e];
  norm[0] = (float)sqrt(acc);
  coef[0] = fminf(1.f, max_norm / (norm[0] + 1e-6f));
  return 0;
}
