// CPU build of lab4d_amd/csrc/sample_pdf_math.hpp for the test-suite (see fk_host.cpp): the per-ray arithmetic of k_sample_pdf.
// Compile with -ffp-contract=off.  TEST INFRASTRUCTURE ONLY.
#include "sample_pdf_math.hpp"

extern "C" float row_sum_host(const float* w, int n, float eps) {
  return n >= 8 ? lab4d_pdf::torch_cpu_row_sum<8>(w, n, eps) : lab4d_pdf::torch_cpu_row_sum<1>(w, n, eps);
}

extern "C" int sample_pdf_host(const float* bins, const float* weights, const float* u_sorted, int R, int n_w, int n_imp, float eps, float* samples,
                               long long* inds) {
  for (int r = 0; r < R; ++r)
    lab4d_pdf::sample_pdf_ray(bins + (long)r * (n_w + 1), weights + (long)r * n_w, n_w, n_imp, eps, u_sorted ? u_sorted + (long)r * n_imp : nullptr,
                              samples + (long)r * n_imp, (int64_t*)inds + (long)r * n_imp);
  return 0;
}

// the det=True queries of sample_pdf_ray, alone: u_k of torch.linspace(0, 1, n)
extern "C" void linspace01_host(int n, float* out) {
  const float step = 1.0f / (float)(n - 1);
  for (int k = 0; k < n; ++k) out[k] = (k < n / 2) ? lab4d_pdf::mul_rn(step, (float)k) : fmaf(-step, (float)(n - 1 - k), 1.0f);
}
