// Arithmetic of the batch-ingestion gather (include/lab4d_ingest.h) as plain __host__ __device__ C++, so that the CPU suite can
// compile the SAME header with g++ (tests/host_harness/, -ffp-contract=off) and hold it to the reference-generated fixture
// without a GPU.  Everything here is bit-exact by construction: half <-> double conversions are exact / round-to-nearest-even,
// and the bilinear interpolation repeats numpy's fp64 operation order (utils/numpy_utils.py:106-121).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LAB4D_HD __host__ __device__ __forceinline__
#else
#define LAB4D_HD inline
#endif

namespace lab4d_ingest {

// a rounded fp64 product / sum the optimiser cannot contract into an fma (see common.hpp: mul_rn)
LAB4D_HD double dmul(double a, double b) {
  double p = a * b;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(p));
#endif
  return p;
}
LAB4D_HD double dadd(double a, double b) {
  double p = a + b;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(p));
#endif
  return p;
}

LAB4D_HD double bits_to_double(uint64_t u) {
  union { uint64_t u; double d; } c;
  c.u = u;
  return c.d;
}
LAB4D_HD uint64_t double_to_bits(double d) {
  union { uint64_t u; double d; } c;
  c.d = d;
  return c.u;
}
LAB4D_HD float bits_to_float(uint32_t u) {
  union { uint32_t u; float f; } c;
  c.u = u;
  return c.f;
}

// IEEE binary16 -> binary64, exact
LAB4D_HD double half_to_double(uint16_t h) {
  const uint64_t sign = (uint64_t)(h >> 15) << 63;
  const int e = (h >> 10) & 31;
  const uint64_t m = h & 1023u;
  if (e == 31) return bits_to_double(sign | 0x7ff0000000000000ull | (m << 42));  // inf / nan
  if (e == 0) {
    // subnormal: m * 2^-24 (exact in fp64)
    const double v = (double)m * 5.9604644775390625e-08;
    return sign ? -v : v;
  }
  return bits_to_double(sign | ((uint64_t)(e - 15 + 1023) << 52) | (m << 42));
}
// IEEE binary16 -> binary32, exact
LAB4D_HD float half_to_float(uint16_t h) { return (float)half_to_double(h); }

// binary64 -> binary16, round to nearest even, ONE rounding (a detour through fp32 rounds twice and differs at ties)
LAB4D_HD uint16_t double_to_half(double d) {
  const uint64_t u = double_to_bits(d);
  const uint16_t sign = (uint16_t)((u >> 48) & 0x8000u);
  const int e = (int)((u >> 52) & 0x7ff);
  const uint64_t m = u & 0x000fffffffffffffull;
  if (e == 0x7ff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u | (uint16_t)(m >> 42) : 0u));
  const int eh = e - 1023 + 15;  // biased half exponent
  if (eh >= 31) return (uint16_t)(sign | 0x7c00u);  // overflow -> inf
  if (eh <= 0) {
    // subnormal half (or zero): value = 1.m * 2^(e-1023); result mantissa = round(value / 2^-24)
    if (eh < -10) return sign;  // below half of the smallest subnormal
    const uint64_t full = m | 0x0010000000000000ull;  // 53-bit significand
    const int shift = 42 + (1 - eh);                   // bits to drop
    const uint64_t q = full >> shift, rem = full & ((1ull << shift) - 1), half = 1ull << (shift - 1);
    uint64_t r = q;
    if (rem > half || (rem == half && (q & 1))) r += 1;
    return (uint16_t)(sign | (uint16_t)r);  // a carry into bit 10 is the smallest normal: correct as is
  }
  const uint64_t q = m >> 42, rem = m & ((1ull << 42) - 1), half = 1ull << 41;
  uint32_t r = ((uint32_t)eh << 10) | (uint32_t)q;
  if (rem > half || (rem == half && (q & 1))) r += 1;  // a mantissa carry bumps the exponent (up to inf): correct as is
  return (uint16_t)(sign | (uint16_t)r);
}

// numpy_utils.py:106-121 for ONE channel.  feat: (FR, FR, FC) map of the frame in the cache dtype, c: channel.
// xy_loc = rand_xy / H * FR in fp64 (vidloader.py:338: both coordinates are divided by img_size[0]).
template <bool F16>
LAB4D_HD float bilinear_channel(const void* feat, int FR, int FC, int c, int px, int py, int H) {
  const double lx = dmul((double)px / (double)H, (double)FR), ly = dmul((double)py / (double)H, (double)FR);
  // ul = floor(xy).astype(int); frac before the clip, clip to [0, FR-2] after   (:108-111; the reference hard-codes 110 = 112 - 2)
  const double flx = __builtin_floor(lx), fly = __builtin_floor(ly);
  const double x = lx - flx, y = ly - fly;
  int ux = (int)flx, uy = (int)fly;
  ux = ux < 0 ? 0 : (ux > FR - 2 ? FR - 2 : ux);
  uy = uy < 0 ? 0 : (uy > FR - 2 ? FR - 2 : uy);
  auto at = [&](int yy, int xx) -> double {
    const long i = ((long)yy * FR + xx) * FC + c;
    if (F16) return half_to_double(((const uint16_t*)feat)[i]);
    return (double)((const float*)feat)[i];
  };
  const double q11 = at(uy, ux), q12 = at(uy, ux + 1), q21 = at(uy + 1, ux), q22 = at(uy + 1, ux + 1);
  const double omx = 1.0 - x, omy = 1.0 - y;
  // q11*(1-x)*(1-y) + q21*(1-x)*(y-0) + q12*(x-0)*(1-y) + q22*(x-0)*(y-0), evaluated left to right   (:116-121)
  double v = dmul(dmul(q11, omx), omy);
  v = dadd(v, dmul(dmul(q21, omx), y));
  v = dadd(v, dmul(dmul(q12, x), omy));
  v = dadd(v, dmul(dmul(q22, x), y));
  if (F16) return half_to_float(double_to_half(v));  // .astype(float16) then .astype(float32)   (:122, vidloader.py:339)
  return (float)v;
}

}  // namespace lab4d_ingest
