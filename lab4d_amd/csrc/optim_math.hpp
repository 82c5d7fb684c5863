// AdamW update of one element, written the way torch.optim.AdamW's single-tensor path orders it (torch/optim/adamw.py
// _single_tensor_adamw: decay, lerp of the first moment, mul/addcmul of the second, bias corrections, addcdiv), which is
// the optimizer the reference builds (lab4d/engine/trainer.py:185-190: betas (0.9, 0.999), weight_decay 1e-4, one group per
// parameter with its own OneCycleLR learning rate).  Plain C++ (see fk_math.hpp for the LAB4D_HD convention): the CPU
// test-suite compiles this header with g++ and checks it against torch.optim.AdamW.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define LAB4D_HD __host__ __device__ inline
#else
#define LAB4D_HD inline
#endif

namespace lab4d_optim {

struct AdamWHyper {
    float one_minus_beta1, beta2, one_minus_beta2, eps, weight_decay;
    float bc1;       // 1 - beta1^step
    float bc2_sqrt;  // sqrt(1 - beta2^step)
};

LAB4D_HD void adamw_update(float& p, float g, float& m, float& v, float lr, const AdamWHyper& h) {
    p = p * (1.f - lr * h.weight_decay);
    m = m + (g - m) * h.one_minus_beta1;
    v = v * h.beta2 + g * g * h.one_minus_beta2;
    const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
    p = p - (lr / h.bc1) * (m / denom);
}

// segment of element e: first s with e < seg_end[s]  (seg_end ascending, last = n)
LAB4D_HD int segment_of(const long long* seg_end, int nseg, long long e) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (e < seg_end[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

}  // namespace lab4d_optim
