// Valid-sample compaction of the evaluation path -- NeRF.get_valid_idx + NeRF.query_nerf (nnutils/nerf.py:495-528, 769-819;
// geom_utils.py:409-422, 506-517).  The reference builds a bool mask with a dozen element-wise launches, counts it on the host
// (`valid_idx.sum() == 0`), gathers the valid samples with boolean indexing (a second host round trip), runs the field on them and
// scatters into zeros.  Here: one kernel forms the mask, a three-kernel stream compaction (wave ballot + popcount, block counts,
// one-block scan, wave-prefix write) produces the index list AND leaves the count on the device, and the consumers (the chain
// kernels' device-side sample count, gather / scatter below) read it there -- no host synchronisation, so the whole evaluation
// pass can be captured in a hipGraph.  Contract: include/lab4d_hip.h section 2b.
#include "common.hpp"  // (declares the entry points through lab4d_hip.h)

namespace lab4d {

constexpr int kCompactBlock = 1024;  // elements per block of the two mask passes (4 waves x 4 ballots)

// strict inequalities on every axis (check_inside_aabb, geom_utils.py:506-517)
__device__ __forceinline__ bool inside(const float* p, const float* lo, const float* hi) {
  return p[0] > lo[0] && p[0] < hi[0] && p[1] > lo[1] && p[1] < hi[1] && p[2] > lo[2] && p[2] < hi[2];
}

__global__ void __launch_bounds__(256) k_valid_mask(const float* __restrict__ xyz, const float* __restrict__ xyz_t, const float* __restrict__ aabb,
                                                     const float* __restrict__ t_aabb, long S, unsigned char* __restrict__ mask) {
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (long)gridDim.x * blockDim.x) {
    const float p[3] = {xyz[3 * s], xyz[3 * s + 1], xyz[3 * s + 2]};
    bool v = inside(p, aabb, aabb + 3);
    if (v && t_aabb) {
      const float q[3] = {xyz_t[3 * s], xyz_t[3 * s + 1], xyz_t[3 * s + 2]};
      v = inside(q, t_aabb, t_aabb + 3);
    }
    mask[s] = v ? 1 : 0;
  }
}

__global__ void __launch_bounds__(256) k_compact_count(const unsigned char* __restrict__ mask, long S, int* __restrict__ block_count) {
  const long base = (long)blockIdx.x * kCompactBlock;
  int c = 0;
#pragma unroll
  for (int i = 0; i < kCompactBlock / 256; ++i) {
    const long s = base + i * 256 + threadIdx.x;
    c += __popcll(__ballot(s < S && mask[s] != 0));  // same value in every lane of the wave
  }
  __shared__ int w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

// exclusive scan of the block counts by ONE workgroup (nblocks <= a few thousand); total -> *count
__global__ void __launch_bounds__(1024) k_compact_scan(int* __restrict__ block_count, int nblocks, int* __restrict__ count) {
  __shared__ int part[1024];
  const int per = (nblocks + 1023) / 1024;
  const int b0 = threadIdx.x * per;
  int s = 0;
  for (int i = 0; i < per; ++i)
    if (b0 + i < nblocks) s += block_count[b0 + i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
    const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (int i = 0; i < per; ++i)
    if (b0 + i < nblocks) {
      const int c = block_count[b0 + i];
      block_count[b0 + i] = run;
      run += c;
    }
  if (threadIdx.x == 1023) *count = part[1023];
}

__global__ void __launch_bounds__(256) k_compact_write(const unsigned char* __restrict__ mask, long S, const int* __restrict__ block_offset,
                                                        int* __restrict__ idx) {
  const long base = (long)blockIdx.x * kCompactBlock;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __shared__ int wcount[kCompactBlock / 256][4];
  unsigned long long bal[kCompactBlock / 256];
#pragma unroll
  for (int i = 0; i < kCompactBlock / 256; ++i) {
    const long s = base + i * 256 + threadIdx.x;
    bal[i] = __ballot(s < S && mask[s] != 0);
    if (lane == 0) wcount[i][wid] = __popcll(bal[i]);
  }
  __syncthreads();
  int off = block_offset[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kCompactBlock / 256; ++i) {
    // order inside the block: row i (256 consecutive samples), then wave, then lane == ascending sample index
    int before = 0;
    for (int w = 0; w < wid; ++w) before += wcount[i][w];
    const long s = base + i * 256 + threadIdx.x;
    if ((bal[i] >> lane) & 1ull) idx[off + before + __popcll(bal[i] & ((1ull << lane) - 1ull))] = (int)s;
    off += wcount[i][0] + wcount[i][1] + wcount[i][2] + wcount[i][3];
  }
}

// dst[j] = src[idx[j]] for j < *count, zeros for *count <= j < n_rows (so that a consumer sized for n_rows reads finite values)
__global__ void __launch_bounds__(256) k_gather_rows(const float* __restrict__ src, const int* __restrict__ idx, const int* __restrict__ count,
                                                      long n_rows, int C, float* __restrict__ dst) {
  const int n = *count;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_rows * C; e += (long)gridDim.x * blockDim.x) {
    const long j = e / C;
    const int c = (int)(e - j * C);
    dst[e] = j < n ? src[(long)idx[j] * C + c] : 0.f;
  }
}
// dst[idx[j]] = src[j] for j < *count (dst pre-filled by the caller: zeros in query_nerf, nerf.py:812-816)
__global__ void __launch_bounds__(256) k_scatter_rows(const float* __restrict__ src, const int* __restrict__ idx, const int* __restrict__ count,
                                                       long n_rows, int C, float* __restrict__ dst) {
  const long n = *count < n_rows ? *count : n_rows;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n * C; e += (long)gridDim.x * blockDim.x) {
    const long j = e / C;
    const int c = (int)(e - j * C);
    dst[(long)idx[j] * C + c] = src[e];
  }
}
// frame of compacted sample j: idx[j] / spf (0 beyond the count)
__global__ void __launch_bounds__(256) k_frame_of(const int* __restrict__ idx, const int* __restrict__ count, long n_rows, int spf, int* __restrict__ frame) {
  const int n = *count;
  for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n_rows; j += (long)gridDim.x * blockDim.x) frame[j] = j < n ? idx[j] / spf : 0;
}

}  // namespace lab4d
using namespace lab4d;

static int grid_for(long n) {
  long g = (n + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

extern "C" int lab4d_valid_mask(const float* xyz, const float* xyz_t, const float* aabb, const float* t_aabb, long S, unsigned char* mask, void* stream) {
  LAB4D_REQUIRE(xyz && aabb && mask && (t_aabb == nullptr || xyz_t), "valid_mask: null pointer");
  if (S == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_valid_mask, dim3(grid_for(S)), dim3(256), 0, (hipStream_t)stream, xyz, xyz_t, aabb, t_aabb, S, mask);
  return check_launch("valid_mask");
}

extern "C" long lab4d_compact_work_ints(long S) { return (S + kCompactBlock - 1) / kCompactBlock + 1; }

extern "C" int lab4d_compact(const unsigned char* mask, long S, int* idx, int* count, int* work, void* stream) {
  LAB4D_REQUIRE(mask && idx && count && work, "compact: null pointer");
  LAB4D_REQUIRE(S >= 0 && S < (1l << 31), "compact: S out of range");
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)((S + kCompactBlock - 1) / kCompactBlock);
  if (nb == 0) {
    if (int e = zero_async(count, sizeof(int), st)) return e;
    return LAB4D_OK;
  }
  hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(256), 0, st, mask, S, work);
  hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, st, work, nb, count);
  hipLaunchKernelGGL(k_compact_write, dim3(nb), dim3(256), 0, st, mask, S, work, idx);
  return check_launch("compact");
}

extern "C" int lab4d_gather_rows(const float* src, const int* idx, const int* count, long n_rows, int C, float* dst, void* stream) {
  LAB4D_REQUIRE(src && idx && count && dst && C > 0, "gather_rows: bad arguments");
  if (n_rows == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(n_rows * C)), dim3(256), 0, (hipStream_t)stream, src, idx, count, n_rows, C, dst);
  return check_launch("gather_rows");
}

extern "C" int lab4d_scatter_rows(const float* src, const int* idx, const int* count, long n_rows, int C, float* dst, void* stream) {
  LAB4D_REQUIRE(src && idx && count && dst && C > 0, "scatter_rows: bad arguments");
  if (n_rows == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_scatter_rows, dim3(grid_for(n_rows * C)), dim3(256), 0, (hipStream_t)stream, src, idx, count, n_rows, C, dst);
  return check_launch("scatter_rows");
}

extern "C" int lab4d_frame_of(const int* idx, const int* count, long n_rows, int spf, int* frame, void* stream) {
  LAB4D_REQUIRE(idx && count && frame && spf > 0, "frame_of: bad arguments");
  if (n_rows == 0) return LAB4D_OK;
  hipLaunchKernelGGL(k_frame_of, dim3(grid_for(n_rows)), dim3(256), 0, (hipStream_t)stream, idx, count, n_rows, spf, frame);
  return check_launch("frame_of");
}
