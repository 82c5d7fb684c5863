// MultiFields.compose_fields on gfx950 (lab4d/nnutils/multifields.py:339-398): the per-ray samples of two fields (fg, bg)
// are concatenated along the depth axis, z-sorted, and every per-sample key is gathered with the same permutation.
// HBM-bound index work: one pass computes the permutation of each ray, one gather per key moves (R, D, C) rows.
// Contract: include/lab4d_hip.h (section 3b).
#include "common.hpp"

namespace lab4d {

// order[r][p] = index (into the concatenation [a | b]) of the p-th smallest depth of ray r; pos = its inverse.
// Rank by counting: rank(i) = #{j : d_j < d_i} + #{j < i : d_j == d_i}  -- a STABLE sort (ties keep concatenation
// order: fg before bg, and sample order inside a field), which is what torch.argsort returns on the reference's CPU path
// and what its CUDA path returns for distinct keys.  One 256-thread block per ray, depths staged in LDS; Dt <= 1024.
__global__ void __launch_bounds__(256) k_compose_order(const float* __restrict__ da, int Da, const float* __restrict__ db, int Db, int R,
                                                        int* __restrict__ order, int* __restrict__ pos) {
  extern __shared__ float sd[];
  const int Dt = Da + Db;
  for (int r = blockIdx.x; r < R; r += gridDim.x) {
    for (int i = threadIdx.x; i < Dt; i += blockDim.x) sd[i] = i < Da ? da[(size_t)r * Da + i] : db[(size_t)r * Db + (i - Da)];
    __syncthreads();
    for (int i = threadIdx.x; i < Dt; i += blockDim.x) {
      const float d = sd[i];
      int rank = 0;
      // total order with NaN last (torch.sort's convention): without it every NaN depth ranks 0, slots of `order` stay unwritten
      // and the gathers read wild indices -- a NaN (e.g. weights gone NaN upstream) must stay a NaN, not become a memory fault
      const bool dn = d != d;
      for (int j = 0; j < Dt; ++j) {
        const float e = sd[j];
        const bool en = e != e;
        const bool lt = dn ? !en : (e < d);
        const bool eq = dn ? en : (e == d);
        rank += lt || (eq && j < i);
      }
      order[(size_t)r * Dt + rank] = i;
      pos[(size_t)r * Dt + i] = rank;
    }
    __syncthreads();
  }
}

// out[r][p][:] = src[r][idx[r][p]][:], src = virtual concatenation [a (R,Da,C) | b (R,Db,C)] along the depth axis; a NULL
// part reads as zeros (a key one field does not produce, multifields.py:383-389).  idx rows have stride idx_ld.
__global__ void __launch_bounds__(256) k_compose_gather(const float* __restrict__ a, int Da, const float* __restrict__ b, int Db,
                                                         const int* __restrict__ idx, int idx_ld, int R, int Dn, int C, float* __restrict__ out) {
  const long total = (long)R * Dn * C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const long rp = e / C;
    const int p = (int)(rp % Dn);
    const long r = rp / Dn;
    const int i = idx[r * idx_ld + p];
    float v = 0.f;
    if (i < Da) {
      if (a) v = a[((size_t)r * Da + i) * C + c];
    } else if (b) {
      v = b[((size_t)r * Db + (i - Da)) * C + c];
    }
    out[e] = v;
  }
}

}  // namespace lab4d

using namespace lab4d;

extern "C" int lab4d_compose_order(const float* depth_a, int Da, const float* depth_b, int Db, int R, int32_t* order, int32_t* pos, void* stream) {
  LAB4D_REQUIRE(depth_a && depth_b && order && pos, "compose_order: null pointer");
  LAB4D_REQUIRE(Da > 0 && Db > 0 && Da + Db <= 1024, "compose_order: bad depths Da=%d Db=%d (Da+Db <= 1024)", Da, Db);
  if (R == 0) return LAB4D_OK;
  const int grid = R < 16384 ? R : 16384;
  hipLaunchKernelGGL(k_compose_order, dim3(grid), dim3(256), (Da + Db) * sizeof(float), (hipStream_t)stream, depth_a, Da, depth_b, Db, R, order, pos);
  return check_launch("compose_order");
}

extern "C" int lab4d_compose_gather(const float* a, int Da, const float* b, int Db, const int32_t* idx, int idx_ld, int R, int Dn, int C,
                                    float* out, void* stream) {
  LAB4D_REQUIRE(idx && out, "compose_gather: null pointer");
  LAB4D_REQUIRE(Da >= 0 && Db >= 0 && Dn > 0 && C > 0 && idx_ld >= Dn, "compose_gather: bad sizes");
  if (R == 0) return LAB4D_OK;
  const long total = (long)R * Dn * C;
  long g = (total + 255) / 256;
  if (g > 65536) g = 65536;
  hipLaunchKernelGGL(k_compose_gather, dim3((int)g), dim3(256), 0, (hipStream_t)stream, a, Da, b, Db, idx, idx_ld, R, Dn, C, out);
  return check_launch("compose_gather");
}
