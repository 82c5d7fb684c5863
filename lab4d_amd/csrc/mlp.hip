// Host API of the fused MLP stack (include/lab4d_mlp.h) + weight-gradient GEMM + weight packing.
// The chain kernels live in mlp_kernels.hpp / mlp_inst_*.hip.
#include <cstdlib>
#include "mlp_kernels.hpp"
#include "mlp_fused_bwd.hpp"

namespace lab4d {

// =================================================================================================
// weight gradient: dW[o][k] += sum_s dz[o][s] X[k][s]   (both operands stored [block][feature][64 samples])
// =================================================================================================
// A workgroup (4 waves) owns one (TM x 4)-tile block of dW and one chunk of samples.  The 4 waves take the
// 16-sample (bf16) / 8-sample (fp32) steps of the chunk round-robin, so together they consume every 128-byte line
// exactly once; each wave keeps its TM x 4 accumulator tiles in registers, prefetches the operands of its next
// step while the MFMAs of the current one run, and at the end the 4 partial blocks are summed through LDS
// (ds_add_f32) so that only one set of global atomics per workgroup is issued.
// Where a weight-gradient element goes.  Default (cmap == NULL): the kernel-layout (mout_pad, K) scratch matrix.  Mapped: straight
// into a gradient buffer in the REFERENCE layout (mout, ld) -- kernel column k is reference column cmap[k] (or -1: a padding /
// conditioning slot that has no weight there), rows >= mout are padding -- so the caller needs no scatter pass and can hand in
// the parameter's accumulated .grad itself (the adds are atomic).  db_rows: bias rows that exist in the db buffer.
struct DwMap {
  const int* cmap;
  int ld, mout, db_rows;
};
// destination column of kernel column k (resolved ONCE per column a lane owns: a lookup per accumulator element made the 256x256
// ring kernel 39 % slower) ...
__device__ __forceinline__ int dw_col(int k, const DwMap& wm) { return wm.cmap ? wm.cmap[k] : k; }
// ... and the add of one element of row o into that column
__device__ __forceinline__ void dw_add(float* dW, int o, int c, int K, const DwMap& wm, float v) {
  if (wm.cmap) {
    if (c >= 0 && o < wm.mout) atomicAdd(dW + (size_t)o * wm.ld + c, v);
  } else {
    atomicAdd(dW + (size_t)o * K + c, v);
  }
}

template <class P, int TM>
__global__ void __launch_bounds__(256) k_mlp_wgrad(const typename P::store_t* __restrict__ dz, const typename P::store_t* __restrict__ emb,
                                                    const typename P::store_t* __restrict__ actp, int mo_tiles, int ke, int kin,
                                                    int S_pad, int chunk, int spf, int cpf, float* __restrict__ dW, float* __restrict__ db, DwMap wm) {
  constexpr int TN = 4;
  constexpr int SPS = P::BF16 ? 16 : 8;  // samples per step
  __shared__ float red[TM * TN * 16 * 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, row = lane & 31, h = lane >> 5;
  const int K = ke + kin, nk_tiles = K / 32;
  const int ob_n = (mo_tiles + TM - 1) / TM, kb_n = (nk_tiles + TN - 1) / TN;
  const int job = blockIdx.x;
  const int c = job / (ob_n * kb_n), rem = job - c * (ob_n * kb_n);
  const int ob = rem / kb_n, kb = rem - ob * kb_n;
  // cpf > 0: chunks are laid out per frame (cpf chunks per frame) and `db` is the per-frame bias gradient (M, mo_pad)
  int s_begin, s_end;
  if (cpf > 0) {
    const int m = c / cpf, lc = c - m * cpf;
    s_begin = m * spf + lc * chunk;
    s_end = min(min(S_pad, (m + 1) * spf), s_begin + chunk);
    if (db) db += (size_t)m * (mo_tiles * 32);
    // the last frame also owns the zero-padded tail samples
    if (s_end == (m + 1) * spf && S_pad - s_end < 64 && S_pad > s_end) s_end = S_pad;
  } else {
    s_begin = c * chunk;
    s_end = min(S_pad, s_begin + chunk);
  }

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float rs[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) rs[i] = 0.f;

  const typename P::store_t* ap[TM];
  const typename P::store_t* bp[TN];
  bool av_[TM], bv_[TN];
  int bF[TN];  // feature rows of the buffer each B operand lives in (blocked layout stride)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int t = ob * TM + i;
    av_[i] = t < mo_tiles;
    ap[i] = dz + (size_t)(32 * (av_[i] ? t : 0) + row) * 64;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int t = kb * TN + j;
    bv_[j] = t < nk_tiles;
    const int kr = 32 * (bv_[j] ? t : 0) + row;
    bp[j] = kr < ke ? emb + (size_t)kr * 64 : actp + (size_t)(kr - ke) * 64;
    bF[j] = kr < ke ? ke : kin;
  }
  const int off = P::BF16 ? 8 * h : 4 * h;
  const int moF = mo_tiles * 32;
  auto load_step = [&](int s, uint4 (&a4)[TM], uint4 (&b4)[TN]) {
    // blocked layout: sample s of feature row f lives at (s/64)*block_stride(F) + f*64 + s%64
    const size_t blk = (size_t)(s >> 6);
    const int in = (s & 63) + off;
#pragma unroll
    for (int i = 0; i < TM; ++i) a4[i] = av_[i] ? *reinterpret_cast<const uint4*>(ap[i] + blk * block_stride(moF) + in) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) b4[j] = bv_[j] ? *reinterpret_cast<const uint4*>(bp[j] + blk * block_stride(bF[j]) + in) : make_uint4(0, 0, 0, 0);
  };
  auto compute = [&](const uint4 (&a4)[TM], const uint4 (&b4)[TN]) {
    if (kb == 0 && db) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (P::BF16) {
          const unsigned int w[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
#pragma unroll
          for (int q = 0; q < 4; ++q) rs[i] += bf2f((unsigned short)(w[q] & 0xffffu)) + bf2f((unsigned short)(w[q] >> 16));
        } else {
          rs[i] += __uint_as_float(a4[i].x) + __uint_as_float(a4[i].y) + __uint_as_float(a4[i].z) + __uint_as_float(a4[i].w);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) mma_unit<P>(acc[i][j], a4[i], b4[j]);
  };
  // software pipeline, depth 1, two statically-named buffers
  const int step = 4 * SPS;
  int s = s_begin + wid * SPS;
  uint4 a0[TM], b0[TN], a1[TM], b1[TN];
  if (s < s_end) load_step(s, a0, b0);
  while (s < s_end) {
    const int s1 = s + step;
    if (s1 < s_end) load_step(s1, a1, b1);
    compute(a0, b0);
    if (s1 >= s_end) break;
    const int s2 = s1 + step;
    if (s2 < s_end) load_step(s2, a0, b0);
    compute(a1, b1);
    s = s2;
  }
  // ---- cross-wave reduction through LDS, then one set of global atomics per workgroup ----
  if (wid == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
  }
  __syncthreads();
  if (wid != 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(&red[((i * TN + j) * 16 + r) * 64 + lane], acc[i][j][r]);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < TM * TN * 16 * 64; e += 256) {
    const int l2 = e & 63, r = (e >> 6) & 15, ij = e >> 10;
    const int i = ij / TN, j = ij - i * TN;
    const int to = ob * TM + i, tk = kb * TN + j;
    if (to < mo_tiles && tk < nk_tiles) {
      const int o = 32 * to + drow(r, l2 >> 5);
      const int k = 32 * tk + (l2 & 31);
      dw_add(dW, o, dw_col(k, wm), K, wm, red[e]);
    }
  }
  if (kb == 0 && db) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float v = rs[i] + __shfl_xor(rs[i], 32, 64);
      if (h == 0 && av_[i] && 32 * (ob * TM + i) + row < wm.db_rows) atomicAdd(db + 32 * (ob * TM + i) + row, v);
    }
  }
}

// Large layers (>= 8 row tiles): a workgroup owns an 8 x 8 block of 32x32 output tiles (256 x 256), its 4 waves are
// arranged 2 x 2 with a 4 x 4 tile sub-block each and ALL walk the same samples in the same order: the two waves of
// a row share their dz tiles and the two of a column share their X tiles through the CU's L1 (same lines requested
// at the same time), which halves the L2 -> L1 traffic of the 4x4-per-wave scheme above; every output tile has a
// single owner, so no cross-wave reduction is needed.
template <class P>
__global__ void __launch_bounds__(256) k_mlp_wgrad_big(const typename P::store_t* __restrict__ dz, const typename P::store_t* __restrict__ emb,
                                                        const typename P::store_t* __restrict__ actp, int mo_tiles, int ke, int kin,
                                                        int S_pad, int chunk, int spf, int cpf, float* __restrict__ dW, float* __restrict__ db, DwMap wm) {
  constexpr int TM = 4, TN = 4;
  constexpr int SPS = P::BF16 ? 16 : 8;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, row = lane & 31, h = lane >> 5;
  const int wr = wid >> 1, wc = wid & 1;
  const int K = ke + kin, nk_tiles = K / 32;
  const int ob_n = (mo_tiles + 7) / 8, kb_n = (nk_tiles + 7) / 8;
  const int job = blockIdx.x;
  const int c = job / (ob_n * kb_n), rem = job - c * (ob_n * kb_n);
  const int ob = rem / kb_n, kb = rem - ob * kb_n;
  int s_begin, s_end;
  if (cpf > 0) {
    const int m = c / cpf, lc = c - m * cpf;
    s_begin = m * spf + lc * chunk;
    s_end = min(min(S_pad, (m + 1) * spf), s_begin + chunk);
    if (db) db += (size_t)m * (mo_tiles * 32);
  } else {
    s_begin = c * chunk;
    s_end = min(S_pad, s_begin + chunk);
  }
  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float rs[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) rs[i] = 0.f;
  const typename P::store_t* ap[TM];
  const typename P::store_t* bp[TN];
  bool av_[TM], bv_[TN];
  int bF[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int t = ob * 8 + wr * 4 + i;
    av_[i] = t < mo_tiles;
    ap[i] = dz + (size_t)(32 * (av_[i] ? t : 0) + row) * 64;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int t = kb * 8 + wc * 4 + j;
    bv_[j] = t < nk_tiles;
    const int kr = 32 * (bv_[j] ? t : 0) + row;
    bp[j] = kr < ke ? emb + (size_t)kr * 64 : actp + (size_t)(kr - ke) * 64;
    bF[j] = kr < ke ? ke : kin;
  }
  const bool any = (av_[0] && bv_[0]);
  const bool do_db = (kb == 0 && wc == 0 && db != nullptr);
  const int off = P::BF16 ? 8 * h : 4 * h;
  const int moF = mo_tiles * 32;
  auto load_step = [&](int s, uint4 (&a4)[TM], uint4 (&b4)[TN]) {
    const size_t blk = (size_t)(s >> 6);
    const int in = (s & 63) + off;
#pragma unroll
    for (int i = 0; i < TM; ++i) a4[i] = av_[i] ? *reinterpret_cast<const uint4*>(ap[i] + blk * block_stride(moF) + in) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) b4[j] = bv_[j] ? *reinterpret_cast<const uint4*>(bp[j] + blk * block_stride(bF[j]) + in) : make_uint4(0, 0, 0, 0);
  };
  auto compute = [&](const uint4 (&a4)[TM], const uint4 (&b4)[TN]) {
    if (do_db) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (P::BF16) {
          const unsigned int w[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
#pragma unroll
          for (int q = 0; q < 4; ++q) rs[i] += bf2f((unsigned short)(w[q] & 0xffffu)) + bf2f((unsigned short)(w[q] >> 16));
        } else {
          rs[i] += __uint_as_float(a4[i].x) + __uint_as_float(a4[i].y) + __uint_as_float(a4[i].z) + __uint_as_float(a4[i].w);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) mma_unit<P>(acc[i][j], a4[i], b4[j]);
  };
  if (any) {
    int s = s_begin;
    uint4 a0[TM], b0[TN], a1[TM], b1[TN];
    if (s < s_end) load_step(s, a0, b0);
    while (s < s_end) {
      const int s1 = s + SPS;
      if (s1 < s_end) load_step(s1, a1, b1);
      compute(a0, b0);
      if (s1 >= s_end) break;
      const int s2 = s1 + SPS;
      if (s2 < s_end) load_step(s2, a0, b0);
      compute(a1, b1);
      s = s2;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (!av_[i]) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!bv_[j]) continue;
        const int c = dw_col(32 * (kb * 8 + wc * 4 + j) + row, wm);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = 32 * (ob * 8 + wr * 4 + i) + drow(r, h);
          dw_add(dW, o, c, K, wm, acc[i][j][r]);
        }
      }
      if (do_db) {
        const float v = rs[i] + __shfl_xor(rs[i], 32, 64);
        if (h == 0 && 32 * (ob * 8 + wr * 4 + i) + row < wm.db_rows) atomicAdd(db + 32 * (ob * 8 + wr * 4 + i) + row, v);
      }
    }
  }
}

// bf16, 256 output rows: LDS-DMA ring.  In the blocked activation layout a 64-sample block of all 256 feature rows is one
// contiguous 32 KiB region; a STAGE is one 32-sample half of such a block for both operands (A = dz, B = X: 2 x 16 KiB),
// filled by 32 lane-linear 1-KiB `global_load_lds_dwordx4` transfers (8 per wave) with no VGPR round trip.  Four stages
// ring through 128 KiB of LDS, so up to three stages (96 KiB per CU) are in flight while the 4 waves (2x2, 4x4 tiles
// each) run 32 MFMAs each on the fourth.  The kernel is HBM-bound (64 KiB of operands per 2048 MFMA-cycles per CU), what
// matters is bytes in flight: the first version (two 64 KiB stages, compiler-managed waits) had NO overlap at all --
// the compiler cannot tell which LDS bytes a DMA writes, so it drained every DMA (s_waitcnt vmcnt(0)) before the first
// ds_read of the other stage.  Here the LDS reads are inline asm (invisible to that analysis), the waits are explicit
// counted vmcnt / lgkmcnt, and the barrier is the raw s_barrier (no fence).
__device__ __forceinline__ void dma_1k(const void* gsrc_lane, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned int*)gsrc_lane,
                                   (__attribute__((address_space(3))) unsigned int*)lds_wave_base, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ u32x4_t lds_read16(unsigned addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MT = row tiles of dz (MO = 32*MT output features: 256 / 128 / 64); NBW = 64-row blocks of X one workgroup contracts
// against (B operand = 64*NBW rows).  Waves 2x2: wave (wr, wc) owns MT/2 x NBW tiles.
// NW = waves per workgroup: 4 (2 x 2 wave grid, TM = MT/2 row tiles per wave) or 8 (4 x 2, TM = MT/4).  The 256-row layers run with
// 8 waves: the same 32-KiB stages in the same LDS ring, but two waves per SIMD (128-160 accumulator registers each instead of
// 256), so the DMA issue, the LDS reads and the barrier wait of one wave overlap the MFMAs of the other, and a 320-column
// operand (the skip layer: embedding + activation) fits ONE job (TN = 5) instead of two that each re-read dZ.
#ifdef LAB4D_ABL_WGRAD4
#define LAB4D_WGRAD_NB5 0
#else
#define LAB4D_WGRAD_NB5 1
#endif
// WC = columns of the wave grid (2: the layers; 4: the <= 32-row heads, MT = 1 -- every wave owns the one row tile and a quarter of
// the 64 NBW columns, the default since round 3; LAB4D_WGRAD_HEAD_DMA=0 selects the pre-DMA kernel).
template <int MT, int NBW, int NW = 4, int WC = 2>
__global__ void __launch_bounds__(64 * NW) k_mlp_wgrad_dma(const unsigned short* __restrict__ dz, const unsigned short* __restrict__ emb,
                                                        const unsigned short* __restrict__ actp, int ke, int kin, int S_pad, int chunk,
                                                        int spf, int cpf, float* __restrict__ dW, float* __restrict__ db, DwMap wm) {
  using P = PBF16;
  static_assert(NW == 4 || NW == 8, "wave grid");
  constexpr int WR = NW / WC;
  static_assert(WR * WC == NW && (2 * NBW) % WC == 0, "wave grid");
  constexpr int TM = MT / WR, TN = 2 * NBW / WC, MO = 32 * MT, KB = 64 * NBW;
  static_assert(TM >= 1 && TM * WR == MT, "row tiles per wave");
  constexpr int A_BYTES = MT * 2048, STAGE = A_BYTES + NBW * 4096;  // 32 samples x (MO + KB) rows x 2 B
  constexpr int NS0 = 65536 / STAGE, NSTAGE = NS0 < 4 ? 4 : (NS0 > 8 ? 8 : NS0);  // ring depth: >= 64 KiB in flight per CU
  constexpr int PA = (2 * MT + NW - 1) / NW, PB = (4 * NBW + NW - 1) / NW, PW = PA + PB;  // transfers per wave per stage (exact: the counted waits rely on it)
  static_assert(PA * NW == 2 * MT || MT == 1, "A pieces per wave (MT = 1: two pieces, the other waves re-fetch the last one)");
  static_assert(PW * (NSTAGE - 2) <= 63, "vmcnt range");
  static_assert(NSTAGE * STAGE <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSTAGE * STAGE];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), row = lane & 31, h = lane >> 5;
  const int wr = wid / WC, wc = wid % WC;
  const int K = ke + kin;
  const int kb_n = (K + KB - 1) / KB;
  const int job = blockIdx.x;
  const int c = job / kb_n, kb = job - c * kb_n;
  int s_begin, s_end;
  if (cpf > 0) {
    const int m = c / cpf, lc = c - m * cpf;
    s_begin = m * spf + lc * chunk;
    s_end = min(min(S_pad, (m + 1) * spf), s_begin + chunk);
    if (db) db += (size_t)m * MO;
  } else {
    s_begin = c * chunk;
    s_end = min(S_pad, s_begin + chunk);
  }
  // B rows [k0, k0+nb) of X = [emb (ke rows) ; act (kin rows)]: n1 rows from emb (from row k0), then the rest from act
  const int k0 = kb * KB, nb = min(KB, K - k0);
  const int n1 = k0 < ke ? min(nb, ke - k0) : 0;
  const int r2 = (k0 > ke ? k0 : ke) - ke;
  bool bv_[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bv_[j] = (wc * TN + j) * 32 < nb;
  const bool do_db = (kb == 0 && wc == 0 && db != nullptr);

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float rs[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) rs[i] = 0.f;

  // LDS image of a stage operand: [rows][64 B]; row r keeps its 16-byte chunk c (4 per row) at slot c ^ ((r >> 2) & 3).
  // The DMA writes lane-linear (a 1-KiB piece = 16 rows, position = lane), so the swizzle is applied to the SOURCE chunk
  // a lane fetches; the 32-row x one-chunk ds_read_b128 pattern below then touches 16 distinct 4-bank slots per 16-lane
  // group (un-swizzled: 86 % of the LDS-active cycles were bank-conflict cycles).
  const int lane_src = (lane >> 2) * 128 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16);  // within a 16-row piece of a block
  // B pieces past the end of a short B operand (nb < KB) re-fetch its last piece: every wave issues exactly PW transfers
  const int nbp = nb / 16;
  auto issue = [&](int st_idx, int buf) {
    int blk = (s_begin >> 6) + (st_idx >> 1);
    const int half = st_idx & 1;
#ifdef LAB4D_ABL_WGRAD_L2  // timing experiment (results wrong): every workgroup keeps re-reading the same 2,048-sample window, so the operands come from the
    blk &= 31;             // L2 instead of HBM -- the matrix-side ceiling of this kernel as a "weight-gradient server" fed through the cache
#endif
    unsigned char* st = lds + buf * STAGE;
    const unsigned char* ga = reinterpret_cast<const unsigned char*>(dz + (size_t)blk * block_stride(MO)) + half * 64 + lane_src;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      int piece = wid + NW * p;
      if constexpr (PA * NW != 2 * MT) piece = piece < 2 * MT ? piece : 2 * MT - 1;
      dma_1k(ga + piece * 2048, st + piece * 1024);
    }
    unsigned char* sb = st + A_BYTES;
    const unsigned char* g1 = reinterpret_cast<const unsigned char*>(emb + (size_t)blk * block_stride(ke) + (size_t)k0 * 64) + half * 64 + lane_src;
    const unsigned char* g2 = reinterpret_cast<const unsigned char*>(actp + (size_t)blk * block_stride(kin) + (size_t)r2 * 64) + half * 64 + lane_src;
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      int q = wid + NW * p;
      q = q < nbp ? q : nbp - 1;
      const int r0 = 16 * q;  // first row of the piece inside the B operand
      const unsigned char* src = r0 < n1 ? g1 + (size_t)r0 * 128 : g2 + (size_t)(r0 - n1) * 128;
      dma_1k(src, sb + q * 1024);
    }
  };
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const unsigned f = (row >> 2) & 3;
  const unsigned la0 = lds_base + (unsigned)((wr * TM * 32 + row) * 64) + ((unsigned)(h ^ f) * 16u);  // sub-step 0: chunk h
  const unsigned lb0 = lds_base + (unsigned)A_BYTES + (unsigned)((wc * TN * 32 + row) * 64) + ((unsigned)(h ^ f) * 16u);
  auto compute = [&](int buf) {
    const unsigned oa = la0 + (unsigned)buf * STAGE, ob = lb0 + (unsigned)buf * STAGE;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      // chunk 2*sub + h -> slot (2*sub + h) ^ f = slot(sub 0) ^ (2*sub): byte address ^ 32
      const unsigned xa = oa ^ (unsigned)(32 * sub), xb = ob ^ (unsigned)(32 * sub);
      u32x4_t a4[TM], b4[TN];
      sfor<0, TM>([&](auto ic) { a4[decltype(ic)::value] = lds_read16<decltype(ic)::value * 2048>(xa); });
      sfor<0, TN>([&](auto jc) { b4[decltype(jc)::value] = lds_read16<decltype(jc)::value * 2048>(xb); });
      // the reads are asm the compiler cannot see through: tie every destination to the wait
      if constexpr (TM == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a4[0]), "+v"(a4[1]), "+v"(a4[2]), "+v"(a4[3]));
      else if constexpr (TM == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a4[0]), "+v"(a4[1]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a4[0]));
      if constexpr (TN == 5) asm volatile("" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]), "+v"(b4[4]));
      else if constexpr (TN == 4) asm volatile("" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]));
      else if constexpr (TN == 3) asm volatile("" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]));
      else if constexpr (TN == 2) asm volatile("" : "+v"(b4[0]), "+v"(b4[1]));
      else asm volatile("" : "+v"(b4[0]));
#pragma unroll
      for (int j = 0; j < TN; ++j)  // column tiles past the end of a short B operand multiply zeros (branch-free)
        if (!bv_[j]) b4[j] = u32x4_t{0u, 0u, 0u, 0u};
      if (do_db) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const unsigned int w[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
#pragma unroll
          for (int q = 0; q < 4; ++q) rs[i] += bf2f((unsigned short)(w[q] & 0xffffu)) + bf2f((unsigned short)(w[q] >> 16));
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bf16x8_t av, bv;
          __builtin_memcpy(&av, &a4[i], 16);
          __builtin_memcpy(&bv, &b4[j], 16);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i][j], 0, 0, 0);
        }
    }
  };
  const int N = ((s_end >> 6) - (s_begin >> 6)) * 2;  // 32-sample stages
  if (N > 0) {
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; ++i)
      if (i < N) issue(i, i);
    int buf = 0;
    for (int s = 0; s < N; ++s) {
      // this wave's PW transfers of stage s have landed once at most the later stages' transfers are outstanding
      const int later = min(NSTAGE - 2, N - 1 - s);
      if (later == NSTAGE - 2) wait_vmcnt<PW * (NSTAGE - 2)>();
      else {
        bool done = false;
        sfor<0, NSTAGE - 2>([&](auto rc) {
          constexpr int R = decltype(rc)::value;
          if (!done && later == R) { wait_vmcnt<PW * R>(); done = true; }
        });
      }
      __builtin_amdgcn_s_barrier();  // everybody's part of stage s has landed AND everybody is done reading stage s-1 ...
      asm volatile("" ::: "memory");
      if (s + NSTAGE - 1 < N) issue(s + NSTAGE - 1, buf == 0 ? NSTAGE - 1 : buf - 1);  // ... whose buffer this stage overwrites
      compute(buf);
      buf = buf + 1 == NSTAGE ? 0 : buf + 1;
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (!bv_[j]) continue;
      const int c = dw_col(k0 + 32 * (wc * TN + j) + row, wm);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = 32 * (wr * TM + i) + drow(r, h);
        dw_add(dW, o, c, K, wm, acc[i][j][r]);
      }
    }
    if (do_db) {
      const float v = rs[i] + __shfl_xor(rs[i], 32, 64);
      if (h == 0 && 32 * (wr * TM + i) + row < wm.db_rows) atomicAdd(db + 32 * (wr * TM + i) + row, v);
    }
  }
}

// per-frame bias gradient: pf_db[m][o] += sum_{s in frame m} dz[o][s].  One wave per (row o, 4096-sample
// segment); a segment that straddles frames flushes at the boundary.  pf_db is zero-filled by the caller.
template <class P>
__global__ void __launch_bounds__(256) k_rowsum_pf(const typename P::store_t* __restrict__ dz, int mo_pad, int S, int ld, int spf,
                                                    int M, int nseg, float* __restrict__ pf_db) {
  constexpr int SEG = 4096;
  const int lane = threadIdx.x & 63;
  const long job = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (job >= (long)mo_pad * nseg) return;
  const int o = (int)(job / nseg), seg = (int)(job % nseg);
  const long b = (long)seg * SEG, e = min((long)S, b + SEG);
  const typename P::store_t* row = dz + (size_t)o * 64;
  long i = b;
  while (i < e) {
    const int m = (int)(i / spf);
    const long fe = min(e, (long)(m + 1) * spf);
    float s = 0.f;
    for (long k = i + lane; k < fe; k += 64) {
      const size_t idx = (size_t)(k >> 6) * block_stride(mo_pad) + (k & 63);  // blocked [block][feature][64] (+skew)
      if constexpr (P::BF16) s += bf2f(row[idx]);
      else s += row[idx];
    }
    s = wave_sum(s);
    if (lane == 0) atomicAdd(pf_db + (size_t)m * mo_pad + o, s);
    i = fe;
  }
}

// =================================================================================================
// weight packing
// =================================================================================================
template <class P>
__global__ void __launch_bounds__(256) k_pack(const float* __restrict__ Wref, int mout, int k_ref, const int32_t* __restrict__ col_map,
                                               int ke, int kin, int mo_pad, int transposed, typename P::store_t* __restrict__ out) {
  constexpr int EPL = P::BF16 ? 8 : 4;  // elements per lane per group
  const int K = ke + kin;
  const int rows = transposed ? K : mo_pad;      // A rows
  const int kdim = transposed ? mo_pad : K;      // contraction length
  const int G = kdim / P::FPG;
  const long total = (long)(rows / 32) * G * 64 * EPL;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e % EPL);
    const int lane = (int)((e / EPL) % 64);
    const int g = (int)((e / (EPL * 64)) % G);
    const int mt = (int)(e / ((long)EPL * 64 * G));
    const int row = 32 * mt + (lane & 31), h = lane >> 5;
    // contraction index -> "natural" index along the contracted dimension
    const int ge = transposed ? 0 : ke / P::FPG;  // identity-ordered groups (embedding block) come first in the forward operand
    int kk;
    if (g < ge) {
      kk = P::BF16 ? 16 * g + 8 * h + j : 2 * (4 * g + j) + h;
    } else {
      const int gg = g - ge;
      int b, r;
      if (P::BF16) { b = gg >> 1; r = 8 * (gg & 1) + j; }
      else { const int kap = 4 * gg + j; b = kap >> 4; r = kap & 15; }
      kk = (transposed ? 0 : ke) + 32 * b + drow(r, h);
    }
    const int o = transposed ? kk : row;   // output feature
    const int c = transposed ? row : kk;   // kernel input column
    float v = 0.f;
    if (o < mout && c < K) {
      const int cr = col_map[c];
      if (cr >= 0) v = Wref[(size_t)o * k_ref + cr];
    }
    if constexpr (P::BF16) out[e] = f2bf(v);
    else out[e] = v;
  }
}

}  // namespace lab4d
using namespace lab4d;

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
namespace {

template <class F>
int with_net(int net, F&& f) {
  switch (net) {
    case LAB4D_NET_FG_BASE: return f(NetFgBase{});
    case LAB4D_NET_FG_COLOR: return f(NetFgColor{});
    case LAB4D_NET_VIS: return f(NetVis{});
    case LAB4D_NET_FEAT: return f(NetFeat{});
    case LAB4D_NET_SKIN: return f(NetSkin{});
    case LAB4D_NET_SKIN18: return f(NetSkin18{});
    case LAB4D_NET_SKIN_A: return f(NetSkinA{});
    case LAB4D_NET_SKIN18_A: return f(NetSkin18A{});
    case LAB4D_NET_DENSE: return f(NetDense{});
    case LAB4D_NET_DENSE6: return f(NetDense6{});
    case LAB4D_NET_BG_BASE: return f(NetBgBase{});
    case LAB4D_NET_BG_COLOR: return f(NetBgColor{});
    case LAB4D_NET_HASH_GEO: return f(NetHashGeo{});
    case LAB4D_NET_HASH_COLOR: return f(NetHashColor{});
    default: set_error("unknown net id %d", net); return LAB4D_EINVAL;
  }
}

}  // namespace

extern "C" int lab4d_mlp_describe(int net, lab4d_mlp_desc* out) {
  LAB4D_REQUIRE(out, "mlp_describe: null out");
  memset(out, 0, sizeof(*out));
  return with_net(net, [&](auto n) { fill_desc<decltype(n)>(out); return LAB4D_OK; });
}

extern "C" int64_t lab4d_mlp_packed_bytes(int net, int layer, int precision) {
  lab4d_mlp_desc d;
  if (lab4d_mlp_describe(net, &d) != LAB4D_OK || layer < 0 || layer >= d.n_layers) return -1;
  const lab4d_mlp_layer& L = d.layers[layer];
  return (int64_t)L.mout_pad * (L.ke + L.kin) * (precision == LAB4D_PREC_BF16 ? 2 : 4);
}

extern "C" int lab4d_mlp_pack(int net, int layer, int precision, int transposed, const float* W_ref, int k_ref,
                              const int32_t* col_map, void* packed, void* stream) {
  lab4d_mlp_desc d;
  if (int e = lab4d_mlp_describe(net, &d)) return e;
  LAB4D_REQUIRE(layer >= 0 && layer < d.n_layers, "mlp_pack: bad layer %d", layer);
  LAB4D_REQUIRE(W_ref && col_map && packed, "mlp_pack: null pointer");
  const lab4d_mlp_layer& L = d.layers[layer];
  const long total = (long)L.mout_pad * (L.ke + L.kin);
  int grid = div_up(total, 256); if (grid > 2048) grid = 2048;
  if (precision == LAB4D_PREC_BF16)
    hipLaunchKernelGGL((k_pack<PBF16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, W_ref, L.mout, k_ref, col_map, L.ke, L.kin,
                       L.mout_pad, transposed, (unsigned short*)packed);
  else if (precision == LAB4D_PREC_F32)
    hipLaunchKernelGGL((k_pack<PF32>), dim3(grid), dim3(256), 0, (hipStream_t)stream, W_ref, L.mout, k_ref, col_map, L.ke, L.kin,
                       L.mout_pad, transposed, (float*)packed);
  else { set_error("mlp_pack: bad precision %d", precision); return LAB4D_EINVAL; }
  return check_launch("mlp_pack");
}

extern "C" int lab4d_mlp_forward(const lab4d_mlp_fwd_args* a, void* stream) {
  LAB4D_REQUIRE(a, "mlp_forward: null args");
  LAB4D_REQUIRE(a->S >= 0 && a->S_pad >= a->S && a->S_pad % 256 == 0 && a->spf > 0 && a->ld >= a->S_pad && a->ld % 8 == 0,
                "mlp_forward: bad sizes (S_pad must be a multiple of 256) S=%d S_pad=%d ld=%d spf=%d", a->S, a->S_pad, a->ld, a->spf);
  LAB4D_REQUIRE(a->x && a->out, "mlp_forward: null x/out");
  if (a->S == 0) return LAB4D_OK;
  return with_net(a->net, [&](auto n) {
    using Net = decltype(n);
    FwdK k;
    memset(&k, 0, sizeof(k));
    k.S = a->S; k.S_pad = a->S_pad; k.ld = a->ld; k.spf = a->spf; k.x = a->x; k.freq_w = a->freq_w; k.emb = a->emb; k.ext = a->ext; k.out = a->out; k.x2 = a->x2;
    k.S_dev = a->S_dev; k.frame_idx = a->frame_idx; k.aff = a->aff;
    LAB4D_REQUIRE(!a->aff || Net::EMB != 0, "mlp_forward: aff is for the raw-input nets only");
    LAB4D_REQUIRE(Net::EMB != 2 || a->aff, "mlp_forward: this network needs the per-frame affine table aff (M, ke, 4)");
    LAB4D_REQUIRE(!Net::AUX3 || a->x2, "mlp_forward: this network needs the second input x2");
    LAB4D_REQUIRE(!(a->S_dev && a->emb), "mlp_forward: a device-side sample count is for the evaluation path (no stored activations)");
    for (int l = 0; l < Net::NL; ++l) {
      LAB4D_REQUIRE(a->W[l] && a->bias[l], "mlp_forward: layer %d weights/bias missing", l);
      LAB4D_REQUIRE(!Net::L[l].pf || a->pf_bias[l], "mlp_forward: layer %d needs a per-frame bias", l);
      LAB4D_REQUIRE(!Net::L[l].add_ext || a->ext, "mlp_forward: layer %d needs ext", l);
      // training mode (emb given): the kernel stores every hidden activation and ReLU mask unconditionally
      if (a->emb && l + 1 < Net::NL) {
        // ... or none of them (point-gradient-only mode: masks + embedding, lab4d_mlp.h)
        // (the layer another net consumes is not exported in that mode either: round 4 found the mode unreachable for the sdf basefields, whose
        // layer 8 / 5 feeds the colour net, because this check still asked for its buffer)
        LAB4D_REQUIRE(a->act[l] || (dx_only_ok<Net>() && !a->act[0]), "mlp_forward: training mode (emb != NULL) needs act[%d]", l);
        LAB4D_REQUIRE(!Net::L[l].relu || a->mask[l], "mlp_forward: training mode (emb != NULL) needs mask[%d]", l);
      }
      k.W[l] = a->W[l]; k.bias[l] = a->bias[l]; k.pf_bias[l] = a->pf_bias[l]; k.act[l] = a->act[l]; k.mask[l] = (unsigned int*)a->mask[l];
    }
    return launch_mlp_fwd<Net>(a->precision, k, a->S, (hipStream_t)stream);
  });
}

extern "C" int lab4d_mlp_backward(const lab4d_mlp_bwd_args* a, void* stream) {
  LAB4D_REQUIRE(a, "mlp_backward: null args");
  LAB4D_REQUIRE(a->S >= 0 && a->S_pad >= a->S && a->S_pad % 256 == 0 && a->spf > 0 && a->ld >= a->S_pad && a->ld % 8 == 0, "mlp_backward: bad sizes (S_pad must be a multiple of 256)");
  LAB4D_REQUIRE(a->d_out, "mlp_backward: null d_out");
  if (a->S == 0) return LAB4D_OK;
  return with_net(a->net, [&](auto n) {
    using Net = decltype(n);
    BwdK k;
    memset(&k, 0, sizeof(k));
    k.S = a->S; k.S_pad = a->S_pad; k.ld = a->ld; k.spf = a->spf; k.emb = a->emb; k.ext = a->ext; k.d_out = a->d_out; k.ext_gin = a->ext_gin;
    k.ext_gout = a->ext_gout; k.d_x = a->d_x; k.d_x2 = a->d_x2; k.x = a->x; k.aff = a->aff; k.g_aff = a->g_aff;
    if (Net::EMB == 2) {
      LAB4D_REQUIRE(a->x && a->aff && a->emb, "mlp_backward: this network needs x, aff and the stored embedding");
    } else {
      LAB4D_REQUIRE(!a->g_aff, "mlp_backward: g_aff is for the affine-first-layer nets only");
    }
    LAB4D_REQUIRE(!(a->d_x && Net::EMB == 0) || a->emb, "mlp_backward: d_x needs the stored embedding");
    for (int l = 0; l < Net::NL; ++l) {
      LAB4D_REQUIRE(a->WT[l], "mlp_backward: layer %d transposed weights missing", l);
      LAB4D_REQUIRE(!(Net::L[l].relu && l + 1 < Net::NL) || a->mask[l], "mlp_backward: layer %d ReLU mask missing", l);
      LAB4D_REQUIRE(!Net::L[l].ext_grad || a->ext_gin, "mlp_backward: layer %d needs ext_gin", l);
      LAB4D_REQUIRE(a->dz[l] || (dx_only_ok<Net>() && !a->dz[0] && a->d_x), "mlp_backward: dz[%d] missing (every layer's dZ is written, or none: point gradient only)", l);
      LAB4D_REQUIRE(!Net::L[l].add_ext || a->ext_gout, "mlp_backward: layer %d needs ext_gout", l);
      k.WT[l] = a->WT[l]; k.act[l] = a->act[l]; k.mask[l] = (const unsigned int*)a->mask[l]; k.dz[l] = a->dz[l];
    }
    return launch_mlp_bwd<Net>(a->precision, k, a->S, (hipStream_t)stream);
  });
}

extern "C" int lab4d_mlp_fused_backward_supported(int net, int precision, int spf) {
  if (precision != LAB4D_PREC_BF16 || spf <= 0 || spf % 64 != 0) return 0;
  // measured per 16.7 M samples (profiles/r04_fused_narrow.json): delta-skin net 5.99 -> 4.65 ms per evaluation (forward + backward + weight gradients);
  // visibility net 5.6 -> 5.9 ms (its posenc is evaluated three times per tile): the fused entry is built and tested for it, and off unless asked for
  static const int vis_env = getenv("LAB4D_FUSED_VIS") ? atoi(getenv("LAB4D_FUSED_VIS")) : 0;
  return net == LAB4D_NET_SKIN_A || net == LAB4D_NET_SKIN18_A || (net == LAB4D_NET_VIS && vis_env != 0);
}

extern "C" int lab4d_mlp_backward_fused(const lab4d_mlp_bwd_fused_args* a, void* stream) {
  LAB4D_REQUIRE(a, "mlp_backward_fused: null args");
  LAB4D_REQUIRE(lab4d_mlp_fused_backward_supported(a->net, a->precision, a->spf), "mlp_backward_fused: net %d / precision %d / spf %d not supported (bf16, narrow nets, spf %% 64 == 0)",
                a->net, a->precision, a->spf);
  LAB4D_REQUIRE(a->S >= 0 && a->x && a->d_out, "mlp_backward_fused: null x / d_out");
  if (a->S == 0) return LAB4D_OK;
  auto run = [&](auto n) {
    using Net = decltype(n);
    FusedK k;
    memset(&k, 0, sizeof(k));
    k.S = a->S; k.spf = a->spf; k.ntiles = (a->S + 63) / 64; k.x = a->x; k.freq_w = a->freq_w; k.aff = a->aff; k.d_out = a->d_out; k.d_x = a->d_x; k.g_aff = a->g_aff;
    LAB4D_REQUIRE(Net::EMB != 2 || a->aff, "mlp_backward_fused: this network needs the per-frame affine table aff");
    for (int l = 0; l < Net::NL; ++l) {
      LAB4D_REQUIRE(a->WT[l] && a->dW[l], "mlp_backward_fused: layer %d transposed weights / dW missing", l);
      LAB4D_REQUIRE(l + 1 == Net::NL || (a->W[l] && (Net::L[l].pf ? (const void*)a->pf_bias[l] : (const void*)a->bias[l])), "mlp_backward_fused: layer %d forward weights / bias missing", l);
      LAB4D_REQUIRE((((uintptr_t)a->W[l] | (uintptr_t)a->WT[l]) & 15) == 0, "mlp_backward_fused: packed weights must be 16-byte aligned");
      k.W[l] = a->W[l]; k.WT[l] = a->WT[l]; k.bias[l] = a->bias[l]; k.pf_bias[l] = a->pf_bias[l]; k.dW[l] = a->dW[l]; k.db[l] = a->db[l]; k.pf_db[l] = a->pf_db[l];
    }
    return launch_mlp_bwd_fused<Net>(k, (hipStream_t)stream);
  };
  switch (a->net) {
    case LAB4D_NET_VIS: return run(NetVis{});
    case LAB4D_NET_SKIN_A: return run(NetSkinA{});
    default: return run(NetSkin18A{});
  }
}

extern "C" int lab4d_mlp_wgrad(int net, int layer, int precision, int S, int S_pad, int ld, int spf, const void* dz, const void* emb,
                               const void* act_prev, float* dW, float* db, float* pf_db, int M, void* stream) {
  return lab4d_mlp_wgrad_mapped(net, layer, precision, S, S_pad, ld, spf, dz, emb, act_prev, dW, 0, nullptr, db, pf_db, M, stream);
}

extern "C" int lab4d_mlp_wgrad_mapped(int net, int layer, int precision, int S, int S_pad, int ld, int spf, const void* dz, const void* emb,
                                      const void* act_prev, float* dW, int ld_ref, const int32_t* col_map, float* db, float* pf_db, int M,
                                      void* stream) {
  lab4d_mlp_desc d;
  if (int e = lab4d_mlp_describe(net, &d)) return e;
  LAB4D_REQUIRE(layer >= 0 && layer < d.n_layers, "mlp_wgrad: bad layer %d", layer);
  const lab4d_mlp_layer& L = d.layers[layer];
  LAB4D_REQUIRE(dz && dW, "mlp_wgrad: null dz/dW");
  LAB4D_REQUIRE(col_map == nullptr || ld_ref > 0, "mlp_wgrad_mapped: a column map needs the reference row stride");
  LAB4D_REQUIRE(L.ke == 0 || emb, "mlp_wgrad: layer %d needs the stored embedding", layer);
  LAB4D_REQUIRE(L.kin == 0 || act_prev, "mlp_wgrad: layer %d needs the previous activation", layer);
  LAB4D_REQUIRE(S_pad % 64 == 0 && S_pad >= S && ld >= S_pad && ld % 8 == 0, "mlp_wgrad: bad S_pad/ld");
  if (S == 0) return LAB4D_OK;
  const int mo_tiles = L.mout_pad / 32, nk_tiles = (L.ke + L.kin) / 32;
  const bool big = mo_tiles >= 8;  // 256-wide layers: 8x8-tile workgroup blocks (k_mlp_wgrad_big / k_mlp_wgrad_dma)
  // bf16 layers of 64 / 128 / 256 output features run the LDS-DMA ring; NBW = 64-row blocks of X per workgroup
  const int Kt = L.ke + L.kin;
  static const int head_dma = getenv("LAB4D_WGRAD_HEAD_DMA") ? atoi(getenv("LAB4D_WGRAD_HEAD_DMA")) : 1;  // round 3: on (heads 23.8 -> 19.7 ms per step); 0 = the pre-DMA kernel
  const bool dma = precision == LAB4D_PREC_BF16 && (mo_tiles == 8 || mo_tiles == 4 || mo_tiles == 2 || (mo_tiles == 1 && head_dma && Kt <= 256));
  // K = 320 (skip layer): one 320-column job with the 8-wave kernel of the 256-row layers, 192 + 128 otherwise
  const int nbw = (dma && mo_tiles == 1) ? (Kt <= 128 ? 2 : 4)  // heads: 1 x 4 wave grid, 128 or 256 columns per workgroup
                  : Kt <= 64 ? 1 : (Kt <= 128 ? 2 : (Kt <= 192 ? 3 : (Kt <= 256 ? 4 : (LAB4D_WGRAD_NB5 && mo_tiles == 8 && Kt <= 320 ? 5 : 3))));
  const int TM = mo_tiles >= 4 ? 4 : (mo_tiles >= 2 ? 2 : 1);
  const int ob_n = dma ? 1 : (big ? div_up(mo_tiles, 8) : div_up(mo_tiles, TM));
  const int kb_n = dma ? div_up(Kt, 64 * nbw) : (big ? div_up(nk_tiles, 8) : div_up(nk_tiles, 4));
  // One resident wave of workgroups (256 CUs x 1 for the 256-row layers and the <= 32-row heads, x 2 for the 64 / 128-row rings), each
  // streaming ONE long sample range: every workgroup ends with an fp32 atomic add of its whole dW block into the same addresses
  // (256 KiB for a 256 x 256 layer), and pays one ring fill; with 1024 workgroups (round 1) those two fixed costs were 11 % of the
  // 256-row launch (0.90 -> 0.80 ms at 4.2 M samples).  Chunks are multiples of 256 samples (one 64-sample step per wave).
  static const int jobs_env = getenv("LAB4D_WGRAD_JOBS") ? atoi(getenv("LAB4D_WGRAD_JOBS")) : 0;  // kernel experiments
  const int jobs_target = jobs_env > 0 ? jobs_env : ((dma && mo_tiles < 8) ? 512 : 256);
  int nchunks = jobs_target / (ob_n * kb_n); if (nchunks < 1) nchunks = 1;
  int chunk = div_up(div_up(S_pad, nchunks), 256) * 256; if (chunk < 1024) chunk = 1024;
  // per-frame bias gradient folded into this kernel when frames are 256-sample aligned: chunks never straddle a
  // frame and the row sums of dz go straight to pf_db (saves a second pass over dz)
  const bool fold_pf = pf_db != nullptr && spf % 256 == 0 && M > 0 && (long)M * spf >= S;
  int cpf = 0;
  if (fold_pf) {
    if (chunk > spf) chunk = spf;
    cpf = div_up(spf, chunk);
    nchunks = cpf * M;
  } else {
    nchunks = div_up(S_pad, chunk);
  }
  const int jobs = ob_n * kb_n * nchunks;
  const dim3 grid(jobs), block(256), block8(512);
  float* db_arg = fold_pf ? pf_db : db;
  // mapped mode: db (when it is the target) is the reference bias (mout entries); the per-frame table keeps its padded rows
  const DwMap wm = {col_map, ld_ref, L.mout, (col_map && !fold_pf) ? L.mout : L.mout_pad};
  hipStream_t st = (hipStream_t)stream;
#define WG(P, TMV) hipLaunchKernelGGL((k_mlp_wgrad<P, TMV>), grid, block, 0, st, (const typename P::store_t*)dz, (const typename P::store_t*)emb, \
                                      (const typename P::store_t*)act_prev, mo_tiles, L.ke, L.kin, S_pad, chunk, spf, cpf, dW, db_arg, wm)
#define WGB(P) hipLaunchKernelGGL((k_mlp_wgrad_big<P>), grid, block, 0, st, (const typename P::store_t*)dz, (const typename P::store_t*)emb, \
                                  (const typename P::store_t*)act_prev, mo_tiles, L.ke, L.kin, S_pad, chunk, spf, cpf, dW, db_arg, wm)
#define WGD(MTV, NBV) hipLaunchKernelGGL((k_mlp_wgrad_dma<MTV, NBV>), grid, block, 0, st, (const unsigned short*)dz, (const unsigned short*)emb, \
                                         (const unsigned short*)act_prev, L.ke, L.kin, S_pad, chunk, spf, cpf, dW, db_arg, wm)
#define WGD8(NBV) hipLaunchKernelGGL((k_mlp_wgrad_dma<8, NBV, 8>), grid, block8, 0, st, (const unsigned short*)dz, (const unsigned short*)emb, \
                                     (const unsigned short*)act_prev, L.ke, L.kin, S_pad, chunk, spf, cpf, dW, db_arg, wm)
#define WGD_NB(MTV) do { if (nbw == 1) WGD(MTV, 1); else if (nbw == 2) WGD(MTV, 2); else if (nbw == 3) WGD(MTV, 3); else WGD(MTV, 4); } while (0)
  if (dma) {
    if (mo_tiles == 8) {
#ifdef LAB4D_ABL_WGRAD4
      WGD_NB(8);
#else
      if (nbw == 1) WGD8(1); else if (nbw == 2) WGD8(2); else if (nbw == 3) WGD8(3); else if (nbw == 4) WGD8(4); else WGD8(5);
#endif
    } else if (mo_tiles == 4) WGD_NB(4); else if (mo_tiles == 2) WGD_NB(2);
    else if (nbw == 2) hipLaunchKernelGGL((k_mlp_wgrad_dma<1, 2, 4, 4>), grid, block, 0, st, (const unsigned short*)dz, (const unsigned short*)emb,
                                          (const unsigned short*)act_prev, L.ke, L.kin, S_pad, chunk, spf, cpf, dW, db_arg, wm);
    else hipLaunchKernelGGL((k_mlp_wgrad_dma<1, 4, 4, 4>), grid, block, 0, st, (const unsigned short*)dz, (const unsigned short*)emb,
                            (const unsigned short*)act_prev, L.ke, L.kin, S_pad, chunk, spf, cpf, dW, db_arg, wm);
  }
#undef WGD_NB
#undef WGD8
#undef WGD
  else if (precision == LAB4D_PREC_BF16) { if (big) WGB(PBF16); else if (TM == 4) WG(PBF16, 4); else if (TM == 2) WG(PBF16, 2); else WG(PBF16, 1); }
  else if (precision == LAB4D_PREC_F32) { if (big) WGB(PF32); else if (TM == 4) WG(PF32, 4); else if (TM == 2) WG(PF32, 2); else WG(PF32, 1); }
  else { set_error("mlp_wgrad: bad precision %d", precision); return LAB4D_EINVAL; }
#undef WGB
#undef WG
  if (int e = check_launch("mlp_wgrad")) return e;
  if (pf_db && !fold_pf) {
    LAB4D_REQUIRE(M > 0 && spf > 0, "mlp_wgrad: pf_db needs M and spf");
    const int nseg = div_up(S, 4096);
    const long jobs2 = (long)L.mout_pad * nseg;
    if (precision == LAB4D_PREC_BF16)
      hipLaunchKernelGGL((k_rowsum_pf<PBF16>), dim3(div_up(jobs2, 4)), dim3(256), 0, st, (const unsigned short*)dz, L.mout_pad, S, ld, spf, M, nseg, pf_db);
    else
      hipLaunchKernelGGL((k_rowsum_pf<PF32>), dim3(div_up(jobs2, 4)), dim3(256), 0, st, (const float*)dz, L.mout_pad, S, ld, spf, M, nseg, pf_db);
    return check_launch("mlp_rowsum_pf");
  }
  return LAB4D_OK;
}

// Tangent-mode forward (eikonal term): x = (S, ke) tangent vector in embedding-slot order, mask[l] = sign bits stored by the
// primal forward, no biases.  Stores the tangent activations (act) and the tangent embedding (emb) for lab4d_mlp_wgrad.
namespace {
template <class Net>
int forward_tangent_impl(const lab4d_mlp_fwd_args* a, void* stream) {
  FwdK k;
  memset(&k, 0, sizeof(k));
  k.S = a->S; k.S_pad = a->S_pad; k.ld = a->ld; k.spf = a->spf; k.x = a->x; k.emb = a->emb; k.out = a->out;
  for (int l = 0; l < Net::NL; ++l) {
    LAB4D_REQUIRE(a->W[l], "mlp_forward_tangent: layer %d weights missing", l);
    LAB4D_REQUIRE(!(Net::L[l].relu) || a->mask[l], "mlp_forward_tangent: layer %d primal ReLU mask missing", l);
    LAB4D_REQUIRE(l + 1 == Net::NL || a->act[l], "mlp_forward_tangent: act[%d] missing", l);
    k.W[l] = a->W[l]; k.bias[l] = a->bias[l] ? a->bias[l] : (const float*)a->W[l]; k.act[l] = a->act[l]; k.mask[l] = (unsigned int*)a->mask[l];
  }
  return launch_mlp_fwd_tangent<Net>(a->precision, k, a->S, (hipStream_t)stream);
}
}  // namespace

extern "C" int lab4d_mlp_forward_tangent(const lab4d_mlp_fwd_args* a, void* stream) {
  LAB4D_REQUIRE(a, "mlp_forward_tangent: null args");
  LAB4D_REQUIRE(a->net == LAB4D_NET_FG_BASE || a->net == LAB4D_NET_BG_BASE,
                "mlp_forward_tangent: only the basefield/sdf networks (fg, bg) have an eikonal term (got net %d)", a->net);
  LAB4D_REQUIRE(a->S >= 0 && a->S_pad >= a->S && a->S_pad % 256 == 0 && a->spf > 0, "mlp_forward_tangent: bad sizes (S_pad must be a multiple of 256)");
  LAB4D_REQUIRE(a->x && a->emb, "mlp_forward_tangent: null x / emb");
  if (a->S == 0) return LAB4D_OK;
  if (a->net == LAB4D_NET_FG_BASE) return forward_tangent_impl<NetFgBase>(a, stream);
  return forward_tangent_impl<NetBgBase>(a, stream);
}
