// Multiresolution hash encoding (Mueller et al., "Instant Neural Graphics Primitives", 2022, section 3), one sample at a time.
// BASELINE config 5 names this encoding; the reference has none (lab4d/nnutils/nerf.py:98 is a TODO), so the
// definition below is this repository's restatement of the paper and its parity is UNPINNED against the reference:
//   level l has grid resolution res[l] (host-computed floor(N_min * b^l), b = exp((ln N_max - ln N_min) / (L - 1)));
//   x in [0,1]^3 is scaled by res, the 8 surrounding vertices are looked up -- 1:1 when (res+1)^3 <= T, otherwise through the
//   spatial hash  (ix * 1) ^ (iy * 2654435761) ^ (iz * 805459861)  mod T  (paper eq. 4) -- and blended tri-linearly;
//   F features per level, output (L * F) level-major.  Table layout (L, T, F) fp32.
// Plain C++ (LAB4D_HD convention of fk_math.hpp): the CPU suite builds this header with g++ and holds it to
// oracle/hashgrid_oracle.py; the kernels of hashgrid.hip call the same functions.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define LAB4D_HD __host__ __device__ inline
#else
#define LAB4D_HD inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define LAB4D_ATOMIC_ADD(ptr, v) atomicAdd((ptr), (v))
// x * res must be ROUNDED before the cell origin is subtracted: contracted into fma(x, res, -floor) the fractional position
// differs from the definition by up to ulp(x * res) = 1.2e-4 at res = 2048 (first MI355X run of the large case: 2e-5 off)
// (__fmul_rn alone is still fused by the backend under HIP's default -ffp-contract=fast: the product is made opaque instead)
LAB4D_HD float lab4d_mul_rn(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}
#define LAB4D_MUL_RN(a, b) lab4d_mul_rn((a), (b))
#else
#define LAB4D_ATOMIC_ADD(ptr, v) (*(ptr) += (v))
#define LAB4D_MUL_RN(a, b) ((a) * (b))
#endif

namespace lab4d_hash {

constexpr int MAXF = 8;

LAB4D_HD uint32_t vertex_index(uint32_t ix, uint32_t iy, uint32_t iz, int res, int log2_T) {
    const uint64_t n = (uint64_t)res + 1, T = (uint64_t)1 << log2_T;
    if (n * n * n <= T) return (uint32_t)(ix + n * (iy + n * iz));
    return (ix ^ (iy * 2654435761u) ^ (iz * 805459861u)) & (uint32_t)(T - 1);
}

// cell origin and fractional position of x (clamped to [0,1]) at resolution res; x == 1 lands in the last cell with w = 1
LAB4D_HD void cell_of(const float* x, int res, uint32_t* i0, float* w) {
    for (int a = 0; a < 3; ++a) {
        const float xc = fminf(fmaxf(x[a], 0.f), 1.f);
        const float p = LAB4D_MUL_RN(xc, (float)res);
        float f = floorf(p);
        if (f > (float)(res - 1)) f = (float)(res - 1);
        i0[a] = (uint32_t)f;
        w[a] = p - f;
    }
}

// out[f] = sum over the 8 vertices of weight * table[vertex][f]
LAB4D_HD void encode_level(const float* x, const float* tab, int res, int log2_T, int F, float* out) {
    uint32_t i0[3];
    float w[3];
    cell_of(x, res, i0, w);
    for (int f = 0; f < F; ++f) out[f] = 0.f;
    for (int c = 0; c < 8; ++c) {
        const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
        const float wt = (dx ? w[0] : 1.f - w[0]) * (dy ? w[1] : 1.f - w[1]) * (dz ? w[2] : 1.f - w[2]);
        const float* e = tab + (size_t)vertex_index(i0[0] + dx, i0[1] + dy, i0[2] + dz, res, log2_T) * F;
        for (int f = 0; f < F; ++f) out[f] += wt * e[f];
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
// g_tab[v + f] += val[f] for every lane of the wave, with the lanes of a RUN of equal consecutive vertices combined first (segmented
// inclusive scan over the 64 lanes, one atomic per run and feature from its last lane).  Lanes are consecutive samples of a ray, and a
// ray crosses a coarse cell in ~10 consecutive samples: at 8.4 M samples x 16 levels x 8 vertices x F the fp32 atomics of the first
// version ran into the L2's atomic rate (12 G atomics/s; the table gradient took 11.8 s of a 13 s step).  EVERY lane of the wave must
// call this (lanes without a sample pass val = 0 and any vertex).
__device__ __forceinline__ void wave_run_add(float* g_tab, uint32_t v, const float* val, int F, int lane) {
    const uint32_t prev = (uint32_t)__shfl_up((int)v, 1, 64);
    int flag = (lane == 0) || (prev != v);
    float a[MAXF];
    for (int f = 0; f < F; ++f) a[f] = val[f];
    for (int off = 1; off < 64; off <<= 1) {
        const int of = __shfl_up(flag, off, 64);
        float o[MAXF];
        for (int f = 0; f < F; ++f) o[f] = __shfl_up(a[f], off, 64);
        if (lane >= off && !flag) {
            for (int f = 0; f < F; ++f) a[f] += o[f];
            flag |= of;
        }
    }
    const uint32_t next = (uint32_t)__shfl_down((int)v, 1, 64);
    if (lane == 63 || next != v)
        for (int f = 0; f < F; ++f)
            if (a[f] != 0.f) atomicAdd(g_tab + (size_t)v + f, a[f]);
}
#endif

#if defined(__HIPCC__)
typedef _Float16 lab4d_h2 __attribute__((ext_vector_type(2)));
// scale of the packed-fp16 accumulation from the largest |gradient entry| of the launch (bits of a non-negative float): a power of two that puts the
// largest single contribution at 2^8 -- 8 binades of headroom for a vertex's sum below fp16's 65504, 22 below for the small contributions
__device__ __forceinline__ float h2_scale_of(uint32_t absmax_bits) {
    const float m = __uint_as_float(absmax_bits);
    if (!(m > 0.f) || !(m < 3.0e38f)) return 1.f;
    return exp2f(8.f - ceilf(log2f(m)));
}
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// The same for F = 2 with ONE packed 2 x fp16 atomic per run (global_atomic_pk_add_f16) into a table of 32-bit words (one word = the two features of
// a vertex): the runs are combined in fp32 as above, the run's sum is scaled and rounded to fp16 once.  Why: the fp32 atomics of the table gradient
// run at the L2 channels' atomic rate -- 21 G per second on this part whatever the footprint (1 MiB or 64 MiB), the scope or the data type
// (tools/probes/atomic_scope.hip, profiles/r06_atomic_scope.jsonl) -- so the lever is the NUMBER of atomics, and a packed one carries both features.
__device__ __forceinline__ void wave_run_add_h2(uint32_t* g16, uint32_t vertex, float v0, float v1, float scale, int lane) {
    const uint32_t prev = (uint32_t)__shfl_up((int)vertex, 1, 64);
    int flag = (lane == 0) || (prev != vertex);
    float a0 = v0, a1 = v1;
    for (int off = 1; off < 64; off <<= 1) {
        const int of = __shfl_up(flag, off, 64);
        const float o0 = __shfl_up(a0, off, 64), o1 = __shfl_up(a1, off, 64);
        if (lane >= off && !flag) {
            a0 += o0;
            a1 += o1;
            flag |= of;
        }
    }
    const uint32_t next = (uint32_t)__shfl_down((int)vertex, 1, 64);
    if ((lane == 63 || next != vertex) && (a0 != 0.f || a1 != 0.f)) {
        const lab4d_h2 h = {(_Float16)(a0 * scale), (_Float16)(a1 * scale)};
        __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) lab4d_h2*)(g16 + vertex), h);
    }
}
#endif

// adjoint: g_tab[vertex][f] += weight * g[f] (atomic on the device); gx[a] += d out / d x_a . g  (gx may be null)
// WAVE = true (device only): the whole wave calls this together and the table updates go through wave_run_add; with g16 != NULL (F = 2) through
// wave_run_add_h2 into the level's slab of packed words instead
template <bool WAVE = false>
LAB4D_HD void encode_level_bwd(const float* x, const float* tab, int res, int log2_T, int F, const float* g, float* g_tab, float* gx, int lane = 0,
                               uint32_t* g16 = nullptr, float scale = 1.f) {
    uint32_t i0[3];
    float w[3];
    cell_of(x, res, i0, w);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < 8; ++c) {
        const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
        const float wx = dx ? w[0] : 1.f - w[0], wy = dy ? w[1] : 1.f - w[1], wz = dz ? w[2] : 1.f - w[2];
        const size_t v = (size_t)vertex_index(i0[0] + dx, i0[1] + dy, i0[2] + dz, res, log2_T) * F;
        float dot = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
        if (WAVE) {
            if (g16) {
                const float wt = wx * wy * wz;
                wave_run_add_h2(g16, (uint32_t)(v / 2), wt * g[0], wt * g[1], scale, lane);
            } else if (g_tab) {
                float val[MAXF];
                for (int f = 0; f < F; ++f) val[f] = wx * wy * wz * g[f];
                wave_run_add(g_tab, (uint32_t)v, val, F, lane);
            }
            if (gx)  // (the table is only READ for d/dx: a launch that wants the table gradient alone -- fixed rays -- skips these 8 gathers per level)
                for (int f = 0; f < F; ++f) dot += tab[v + f] * g[f];
        } else
#endif
        for (int f = 0; f < F; ++f) {
            if (g_tab) LAB4D_ATOMIC_ADD(g_tab + v + f, wx * wy * wz * g[f]);
            if (gx) dot += tab[v + f] * g[f];
        }
        acc[0] += (dx ? 1.f : -1.f) * wy * wz * dot;
        acc[1] += wx * (dy ? 1.f : -1.f) * wz * dot;
        acc[2] += wx * wy * (dz ? 1.f : -1.f) * dot;
    }
    if (gx)
        for (int a = 0; a < 3; ++a) gx[a] += acc[a] * (float)res;
}

}  // namespace lab4d_hash
