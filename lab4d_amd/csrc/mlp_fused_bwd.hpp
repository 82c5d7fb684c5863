// Fused backward of the NARROW per-sample networks (every layer <= 64 wide: the visibility field and the affine-form delta-skin fields), bf16.
// Round 4.  Contract: include/lab4d_mlp.h, lab4d_mlp_backward_fused.
//
// These nets are 1-2 % of the step's FLOPs and took 11 % of its time: their training-mode forward stored the embedding, every activation and
// the ReLU masks (384 B per sample and evaluation for the delta-skin net), the backward chain wrote every dZ, and one weight-gradient launch per
// layer read both back -- about 1.3 KB of HBM traffic per sample for 8-10 k MACs.  Here the forward stores NOTHING (it runs in inference mode)
// and ONE backward kernel per evaluation
//   1. recomputes the forward of its 64-sample tile in registers (same arithmetic as k_mlp_fwd: the same packed weights, bias through the MFMA C
//      operand, one fp32 -> bf16 conversion per value),
//   2. runs the dgrad chain on it (masks from the recomputed pre-activations),
//   3. accumulates every layer's weight gradient dW_l += dZ_l X_{l-1}^T in REGISTERS across all tiles of the wave: the two operands sit in the
//      wave's LDS slab as 16-byte B units [n-tile][unit][lane] (8 features of one sample) and come back through ds_read_b64_tr_b16 as
//      [feature][8 samples] MFMA fragments (contraction over samples), 16 MFMAs per 64 x 64 layer and tile,
//   4. leaves dW / db (and the per-frame quantities: bias table rows, the affine first layer's table gradient) with atomics once per wave (per
//      frame for the per-frame ones), after a reduction over the 4 waves of the workgroup in LDS.
// Traffic: 12 B (point) + c_out * 4 B (head gradient) read, 12 B (point gradient) written per sample.  The weights (<= 40 KiB per net, both
// orientations) are resident in LDS for the whole kernel: no weight stream, no step barrier, the four waves of a workgroup never meet.
#pragma once
#include "mlp_kernels.hpp"

namespace lab4d {

constexpr int FUSED_MAXL = 3;  // the narrow nets have <= 3 layers: 27 pointers stay in SGPRs (with 12-entry arrays the 90 pointers were spilled to scratch at the kernel entry)
struct FusedK {
  int S, spf, ntiles;
  const float* x;
  const float* freq_w;
  const float* aff;
  const void* W[FUSED_MAXL];
  const void* WT[FUSED_MAXL];
  const float* bias[FUSED_MAXL];
  const float* pf_bias[FUSED_MAXL];
  const float* d_out;
  float* d_x;
  float* g_aff;
  float* dW[FUSED_MAXL];
  float* db[FUSED_MAXL];
  float* pf_db[FUSED_MAXL];
};

template <class Net>
constexpr bool fused_bwd_ok() {
  if (net_wmax<Net>() > 64 || Net::KE != 64 || Net::NL > FUSED_MAXL || Net::AUX3 || Net::EMB == 1) return false;
  for (int l = 0; l < Net::NL; ++l) {
    const LS s = Net::L[l];
    if ((l == 0) != (s.ke != 0) || (l > 0 && s.kin != 64) || s.add_ext || s.ext_grad || (l + 1 < Net::NL && (!s.relu || s.mout != 64))) return false;
  }
  return pad32(Net::L[Net::NL - 1].mout) == 32 && !Net::L[Net::NL - 1].relu;
}

// LDS byte offsets of this lane's supplier chunk for the transposing fragment reads (see the header comment).  A fragment = rows (features) 0..31 of
// a 32-feature tile `ft`, k = 16 consecutive samples of the 64-sample tile: lane (m = l & 31, kh = l >> 5) receives row m, samples 16 kk + 8 kh + 0..7.
// ds_read_b64_tr_b16: inside a 16-lane group, lane i's element j is element i & 3 of the 8-byte chunk addressed by lane (i >> 2) + 4 j.  With
// c = lane & 15, G = lane >> 4 (gm = G & 1: rows 16 gm .., kh = G >> 1), lane c therefore has to address the chunk that holds rows
// 16 gm + 4 (c & 3) .. + 3 of sample 16 kk + 8 kh + (c >> 2) (second read: + 4).  Slab: unit u of n-tile t at lane (n, h) = 16 bytes,
// sample = 2 n + t.  ACT units (accumulator order): row r of the 32-row tile sits in unit 2 ft + (r >> 4), lane half (r >> 2) & 1, 8-byte chunk
// (r >> 3) & 1; EMB units (identity slot order): slot r in unit 2 ft + (r >> 4), lane half (r >> 3) & 1, chunk (r >> 2) & 1.
// kk adds 8 lanes (128 B), the second read 2 lanes (32 B), ft two units (2048 B): immediates.
template <bool EMB_ORDER>
__device__ __forceinline__ unsigned frag_lane_offset(int lane) {
  const int c = lane & 15, G = lane >> 4, gm = G & 1, kh = G >> 1;
  const int s = 8 * kh + (c >> 2), n = s >> 1, t = s & 1, q4 = c & 3;  // q4: which 4-row group of the 16 rows
  const int half = EMB_ORDER ? (q4 >> 1) : (q4 & 1), chunk = EMB_ORDER ? (q4 & 1) : (q4 >> 1);
  return (unsigned)((((t * 4 + gm) * 64 + n + 32 * half) * 16) + 8 * chunk);
}
struct Frag {
  unsigned long long a, b;  // samples 8 kh + 0..3 / + 4..7 of this lane's row
};
template <int FT, int KK>
__device__ __forceinline__ void frag_issue(unsigned addr, Frag& f) {  // no wait: a layer's fragments are requested back to back (frag_wait4)
  constexpr int OFF = FT * 2048 + KK * 128;
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" : "=&v"(f.a), "=&v"(f.b) : "v"(addr), "n"(OFF), "n"(OFF + 32) : "memory");
}
// everything requested so far has arrived; the fragments are in/out operands so that no consumer can be scheduled in front of the wait
__device__ __forceinline__ void frag_wait4(Frag& f0, Frag& f1, Frag& f2, Frag& f3) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0.a), "+v"(f0.b), "+v"(f1.a), "+v"(f1.b), "+v"(f2.a), "+v"(f2.b), "+v"(f3.a), "+v"(f3.b)::"memory");
}
__device__ __forceinline__ uint4 frag_units(const Frag& f) { return make_uint4((unsigned)f.a, (unsigned)(f.a >> 32), (unsigned)f.b, (unsigned)(f.b >> 32)); }

template <class Net>
__global__ void __launch_bounds__(256, 1) k_mlp_bwd_fused(FusedK a) {
  static_assert(fused_bwd_ok<Net>(), "k_mlp_bwd_fused: narrow nets only");
  using P = PBF16;
  constexpr int NL = Net::NL, NH = NL - 1;  // hidden (64-wide, ReLU) layers 0 .. NH-1, head = layer NH
  constexpr int UW = 4;                     // 16-byte units per n-tile of a 64-wide operand
  // packed weight groups (1 KiB each) resident in LDS: forward operands of the hidden layers (the head's output is not recomputed), transposed
  // operands of every layer
  constexpr int FWD_G = NH * 8;                 // hidden layer: MT = 2 row tiles x G = 4 k-groups
  constexpr int BWD_G = 4 + (NH - 1) * 8 + 8;  // head^T: 2 row tiles x 2 groups; hidden l >= 1: 2 x 4; layer 0: 2 embedding row tiles x 4
  __shared__ uint4 wf[FWD_G * 64];
  __shared__ uint4 wt[BWD_G * 64];
  // per wave NH + 1 slots of [n-tile 2][unit 4][lane 64] 16-byte units: slot l < NH = the post-activation of hidden layer l (written by the forward
  // recompute; slot NH - 1 takes the embedding once the head's weight gradient has consumed it), slot NH = the dZ of the layer being processed
  constexpr int SLOT = 2 * UW * 64, NSLOT = NH + 1;
  __shared__ uint4 slab_all[4 * NSLOT * SLOT];
  __shared__ float4 afftab[Net::EMB == 2 ? 4 * 64 : 1];
  __shared__ float4 biastab[4 * (NH > 0 ? NH : 1) * 16];  // per wave: the 64 biases of every hidden layer (per-frame layers: the current frame's row)
  const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = blockIdx.x * 4 + wid, nwaves = gridDim.x * 4;
  // ---- weights -> LDS (once per workgroup) ----
  {
    int g0 = 0;
    for (int l = 0; l < NH; ++l) {
      const uint4* src = (const uint4*)a.W[l];
      for (int e = threadIdx.x; e < 8 * 64; e += 256) wf[g0 * 64 + e] = src[e];
      g0 += 8;
    }
    g0 = 0;
    for (int l = NL - 1; l >= 0; --l) {
      const int ng = (l == NL - 1) ? 4 : 8;  // groups of layer l's transposed operand that are used: head 2 x 2; hidden 2 x 4 (layer 0: its embedding rows)
      const uint4* src = (const uint4*)a.WT[l];
      for (int e = threadIdx.x; e < ng * 64; e += 256) wt[g0 * 64 + e] = src[e];
      g0 += ng;
    }
  }
  __syncthreads();
  uint4* slab0 = slab_all + wid * (NSLOT * SLOT);  // slot s at slab0 + s * SLOT
  uint4* slabZ = slab0 + NH * SLOT;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)slab0;
  const unsigned ldsZ = lds0 + NH * SLOT * 16;
  const unsigned fo_act = frag_lane_offset<false>(lane), fo_emb = frag_lane_offset<true>(lane);
  float* stagef = reinterpret_cast<float*>(slabZ);  // head-gradient staging: the dZ slot is idle until the head gradient is in registers

  // ---- persistent accumulators: weight gradients (MFMA tiles: rows = dZ features, columns = X features), per-lane bias partial sums ----
  f32x16_t dWh[NH][2][2];  // hidden layers: 64 x 64
  f32x16_t dWo[2];         // head: 32 x 64
  // bias gradients = row sums of dZ: the SAME dZ fragments against an all-ones B fragment (every column of the tile then holds the row sums) --
  // MFMA-updated accumulators (they can live in AGPRs), no VALU adds per tile, no cross-lane reduction at the end
  f32x16_t dbh[NH][2];
  f32x16_t dbo;
  const uint4 ones8 = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);  // eight bf16 1.0
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int l = 0; l < NH; ++l) {
      dWh[l][0][0][r] = dWh[l][0][1][r] = dWh[l][1][0][r] = dWh[l][1][1][r] = 0.f;
      dbh[l][0][r] = dbh[l][1][r] = 0.f;
    }
    dWo[0][r] = dWo[1][r] = dbo[r] = 0.f;
  }
  // EMB == 2: the affine table's gradient g_aff[slot][0..3] = sum_s dz_emb[slot][s] [x_s; 1] is one more contraction over samples: the slot gradients
  // (bf16, like every dZ of this path) against the 8-row operand [x_hi (3) | 1 | x_lo (3) | 0] -- the point split into two bf16 so that it enters
  // with ~16 significant bits -- instead of 128 VALU-updated registers per lane (k_mlp_bwd's gacc)
  f32x16_t gA[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) gA[0][r] = gA[1][r] = 0.f;
  int gframe = -1;
  constexpr bool PF0 = Net::L[0].pf != 0;  // layer 0 takes a per-frame bias table: its bias gradient is per frame

  // reduce the per-lane partials of the per-frame quantities over the 32 lanes of a half and add them to frame `gframe`'s rows
  auto frame_flush = [&]() {
    if (gframe < 0) return;
    if constexpr (Net::EMB == 2) {
      if (a.g_aff != nullptr) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            // lane (n, h) holds column n of row drow(r, h): columns 0..2 = x_hi, 3 = 1, 4..6 = x_lo
            const float v = gA[mt][r], lo = __shfl_down(v, 4, 64);
            if (n < 4) atomicAdd(a.g_aff + ((size_t)gframe * Net::KE + 32 * mt + drow(r, h)) * 4 + n, n < 3 ? v + lo : v);
          }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) gA[0][r] = gA[1][r] = 0.f;
    }
    if constexpr (PF0) {
      if (a.pf_db[0] != nullptr) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (n == 0) atomicAdd(a.pf_db[0] + (size_t)gframe * 64 + 32 * mt + drow(r, h), dbh[0][mt][r]);  // every column holds the row sum
            dbh[0][mt][r] = 0.f;
          }
      }
    }
  };

  for (int tile = wave; tile < a.ntiles; tile += nwaves) {
    const int s0 = tile * 64;
    int sidx[2];
    float xs[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      sidx[t] = s0 + 2 * n + t;
      const int sc = sidx[t] < a.S ? sidx[t] : a.S - 1;
#pragma unroll
      for (int k = 0; k < 3; ++k) xs[t][k] = ((const GLOBAL_AS float*)a.x)[(size_t)sc * 3 + k];
    }
    // every tile lies in one frame (host contract: spf % 64 == 0)
    const int frame = __builtin_amdgcn_readfirstlane((s0 < a.S ? s0 : a.S - 1) / a.spf);
    const bool frame_changed = frame != gframe;
    if (frame_changed) {
      frame_flush();
      gframe = frame;
    }
    float4* btab = biastab + wid * NH * 16;
    if (frame_changed) {
      __builtin_amdgcn_wave_barrier();
      sfor<0, NH>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        const GLOBAL_AS float* bsrc = (Net::L[l].pf != 0) ? (const GLOBAL_AS float*)a.pf_bias[l] + (size_t)frame * 64 : (const GLOBAL_AS float*)a.bias[l];
        if (lane < 16) {
          const f32x4_t v = *(const GLOBAL_AS f32x4_t*)(bsrc + 4 * lane);
          btab[l * 16 + lane] = make_float4(v.x, v.y, v.z, v.w);
        }
      });
      __builtin_amdgcn_wave_barrier();
    }
    float4* ltab = afftab + wid * 64;
    if constexpr (Net::EMB == 2) {
      __builtin_amdgcn_wave_barrier();
      {
        const f32x4_t r4 = *(const GLOBAL_AS f32x4_t*)((const GLOBAL_AS float4*)a.aff + (size_t)frame * 64 + lane);
        ltab[lane] = make_float4(r4.x, r4.y, r4.z, r4.w);
      }
      __builtin_amdgcn_wave_barrier();
    }

    // sin / cos of every band of one sample by angle doubling from one sincos per axis (as k_mlp_fwd's bf16 path)
    float sn0[2][3], cs0[2][3];  // the one accurate sincos per axis and sample; the bands are re-derived from it wherever they are needed
    if constexpr (Net::EMB == 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) sincosf(xs[t][ax], &sn0[t][ax], &cs0[t][ax]);
    }
    auto bands = [&](int t, float (&sv)[Net::NFREQ > 0 ? Net::NFREQ : 1][3], float (&cv)[Net::NFREQ > 0 ? Net::NFREQ : 1][3]) {
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        float sn = sn0[t][ax], cs = cs0[t][ax];
#pragma unroll
        for (int f = 0; f < Net::NFREQ; ++f) {
          sv[f][ax] = sn; cv[f][ax] = cs;
          sincos_double(sn, cs);
        }
      }
    };
    // the embedding of the tile as B units (identity slot order); evaluated twice per tile (forward recompute, layer 0's weight gradient) instead of
    // being held in 32 registers through the whole backward
    auto make_emb = [&](uint4 (&emb)[2][UW]) {
      if constexpr (Net::EMB == 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int g = 0; g < UW; ++g) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float4 r = ltab[16 * g + 8 * h + j];
              v[j] = relu1(r.x * xs[t][0] + r.y * xs[t][1] + r.z * xs[t][2] + r.w);
            }
            emb[t][g] = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
          }
      } else {
        constexpr int L = Net::NFREQ;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float sv[L > 0 ? L : 1][3], cv[L > 0 ? L : 1][3];
          bands(t, sv, cv);
#pragma unroll
          for (int g = 0; g < UW; ++g) {
            unsigned int w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // slots 16g + 8h + 2i, +1 = pair 8g + 4h + i
              float v0[2], v1[2];
#pragma unroll
              for (int hh = 0; hh < 2; ++hh) {
                const int pair = 8 * g + 4 * hh + i;  // compile-time after unrolling
                if (pair < 3 * L) {
                  const int f = pair / 3, ax = pair - 3 * f;
                  const float wf_ = a.freq_w ? a.freq_w[f] : 1.0f;
                  v0[hh] = sv[f < L ? f : 0][ax] * wf_; v1[hh] = cv[f < L ? f : 0][ax] * wf_;
                } else {
                  const int sl = 2 * pair - 6 * L;
                  v0[hh] = sl < 3 ? xs[t][sl < 3 ? sl : 0] : 0.f;
                  v1[hh] = sl + 1 < 3 ? xs[t][sl + 1 < 3 ? sl + 1 : 0] : 0.f;
                }
              }
              w[i] = pack2bf(h ? v0[1] : v0[0], h ? v1[1] : v1[0]);
            }
            emb[t][g] = make_uint4(w[0], w[1], w[2], w[3]);
          }
        }
      }
    };

    // ================= forward recompute: hidden layers 0 .. NH-1, post-activations parked in their slots =================
    unsigned int alive[NH][2];
    {
      uint4 cur[2][UW];  // input units of the layer being evaluated
      make_emb(cur);
      sfor<0, NH>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        uint4 nxt[2][UW];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x16_t acc[2];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 v = btab[l * 16 + 8 * mt + 2 * i + h];  // features 32 mt + 8 i + 4 h + 0..3
            acc[0][4 * i + 0] = v.x; acc[0][4 * i + 1] = v.y; acc[0][4 * i + 2] = v.z; acc[0][4 * i + 3] = v.w;
          }
          acc[1] = acc[0];
#pragma unroll
          for (int g = 0; g < UW; ++g) {
            const uint4 A = wf[((l * 2 + mt) * 4 + g) * 64 + lane];
#pragma unroll
            for (int t = 0; t < 2; ++t) mma_unit<P>(acc[t], A, cur[t][g]);
          }
          unsigned int w[2][8];
#pragma unroll
          for (int t = 0; t < 2; ++t) acc_fence(acc[t]);
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int k = 0; k < 8; ++k) w[t][k] = pack2bf_op(acc[t][2 * k], acc[t][2 * k + 1]);
          alive[l][mt] = pk_alive_bits(w);
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 2; ++q)
              nxt[t][2 * mt + q] = make_uint4(pk_relu_bf16(w[t][4 * q]), pk_relu_bf16(w[t][4 * q + 1]), pk_relu_bf16(w[t][4 * q + 2]), pk_relu_bf16(w[t][4 * q + 3]));
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < UW; ++u) {
            slab0[l * SLOT + (t * UW + u) * 64 + lane] = nxt[t][u];  // X of layer l + 1's weight gradient
            cur[t][u] = nxt[t][u];
          }
      });
    }

    // ================= head gradient =================
    uint4 dzu[2][UW];  // dZ of the current layer as B units (head: units 0, 1 only)
    {
      f32x16_t g[2];
      if constexpr (head_staged<Net>()) {
        __builtin_amdgcn_wave_barrier();
        stage_in<64 * Net::COUT>(stagef, a.d_out, (long)s0 * Net::COUT, (long)a.S * Net::COUT - 1, lane);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int f = drow(r, h);
            g[t][r] = (f < Net::COUT && sidx[t] < a.S) ? stagef[(2 * n + t) * Net::COUT + (f < Net::COUT ? f : 0)] : 0.f;
          }
        __builtin_amdgcn_wave_barrier();
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int f = drow(r, h);
            g[t][r] = (f < Net::COUT && sidx[t] < a.S) ? a.d_out[(size_t)sidx[t] * Net::COUT + f] : 0.f;
          }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dbo[r] += g[0][r] + g[1][r];  // the head's bias gradient: per-lane partial sums (VALU registers; the MFMA accumulators fill the AGPR file)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        uint4 u[2];
        tile_to_units<P>(g[t], u);
        dzu[t][0] = u[0]; dzu[t][1] = u[1];
        dzu[t][2] = dzu[t][3] = make_uint4(0, 0, 0, 0);
      }
    }

    // ================= layers NL-1 .. 0: weight gradient of the layer, then the gradient of its input =================
    float dx[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    sfor<0, NL>([&](auto lc) {
      constexpr int l = NL - 1 - decltype(lc)::value;
      constexpr bool is_head = (l == NL - 1);
      constexpr int GK = is_head ? 2 : 4;  // k-groups of the transposed operand = units of this layer's dZ
      // first LDS group of layer l's transposed operand (loaded head first): head 4 groups, then 8 per hidden layer
      constexpr int wt_g0 = is_head ? 0 : 4 + (NL - 2 - l) * 8;
      // ---- dW_l += dZ_l X_{l-1}^T : dZ_l into its slot; X_{l-1} sits in slot l - 1 since the forward (layer 0: the embedding, evaluated again,
      // into slot NH - 1, which the head's / layer NH-1's weight gradient is done with) ----
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < GK; ++u) slabZ[(t * UW + u) * 64 + lane] = dzu[t][u];
      constexpr int xslot = (l == 0) ? NH - 1 : l - 1;
      if constexpr (l == 0) {
        uint4 emb[2][UW];
        make_emb(emb);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < UW; ++u) slab0[xslot * SLOT + (t * UW + u) * 64 + lane] = emb[t][u];
      }
      __builtin_amdgcn_wave_barrier();
      const unsigned ldsX = lds0 + (unsigned)(xslot * SLOT * 16) + (l == 0 ? fo_emb : fo_act);
      {
        // two k-steps of fragments at a time (4 of X, 2 or 4 of dZ: 12 / 16 transposing reads in flight), then their MFMAs
        sfor<0, 2>([&](auto hc) {
          constexpr int k0 = 2 * decltype(hc)::value;
          Frag fx[2][2], fz[2][2];
          sfor<0, 2>([&](auto kc) {
            constexpr int kq = decltype(kc)::value, kk = k0 + kq;
            frag_issue<0, kk>(ldsX, fx[kq][0]);
            frag_issue<1, kk>(ldsX, fx[kq][1]);
            frag_issue<0, kk>(ldsZ + fo_act, fz[kq][0]);
            if constexpr (!is_head) frag_issue<1, kk>(ldsZ + fo_act, fz[kq][1]);
            else fz[kq][1] = fz[kq][0];
          });
#pragma unroll
          for (int kq = 0; kq < 2; ++kq) frag_wait4(fx[kq][0], fx[kq][1], fz[kq][0], fz[kq][1]);
#pragma unroll
          for (int kq = 0; kq < 2; ++kq) {
            const uint4 fx0 = frag_units(fx[kq][0]), fx1 = frag_units(fx[kq][1]), fz0 = frag_units(fz[kq][0]);
            if constexpr (is_head) {
              mma_unit<P>(dWo[0], fz0, fx0);
              mma_unit<P>(dWo[1], fz0, fx1);
            } else {
              const uint4 fz1 = frag_units(fz[kq][1]);
              mma_unit<P>(dWh[l][0][0], fz0, fx0);
              mma_unit<P>(dWh[l][0][1], fz0, fx1);
              mma_unit<P>(dWh[l][1][0], fz1, fx0);
              mma_unit<P>(dWh[l][1][1], fz1, fx1);
              mma_unit<P>(dbh[l][0], fz0, ones8);
              mma_unit<P>(dbh[l][1], fz1, ones8);
            }
          }
        });
      }
      __builtin_amdgcn_wave_barrier();
      // ---- gradient of the layer's input ----
      if constexpr (l > 0) {
        // activation rows of W_l^T (hidden layers have no embedding rows): dZ_{l-1} = relu'(z_{l-1}) * (W_l^T dZ_l)
        uint4 nz[2][UW];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16_t acc[2];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
#pragma unroll
          for (int g = 0; g < GK; ++g) {
            const uint4 A = wt[(wt_g0 + j * GK + g) * 64 + lane];
#pragma unroll
            for (int t = 0; t < 2; ++t) mma_unit<P>(acc[t], A, dzu[t][g]);
          }
          const unsigned int al = alive[l - 1][j];
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const bool on = ((al >> pk_bit(t, r)) & 1u) != 0u;
              acc[t][r] = on ? acc[t][r] : 0.f;
            }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            uint4 u[2];
            tile_to_units<P>(acc[t], u);
            nz[t][2 * j] = u[0]; nz[t][2 * j + 1] = u[1];
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < UW; ++u) dzu[t][u] = nz[t][u];
      } else {
        if (a.d_x != nullptr) {
          // embedding rows of W_0^T -> point gradient (and, EMB == 2, the affine table's gradient)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            f32x16_t acc[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const uint4 A = wt[(wt_g0 + mt * 4 + g) * 64 + lane];
#pragma unroll
              for (int t = 0; t < 2; ++t) mma_unit<P>(acc[t], A, dzu[t][g]);
            }
            if constexpr (Net::EMB == 0) {
              constexpr int L = Net::NFREQ;
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                float sv[L > 0 ? L : 1][3], cv[L > 0 ? L : 1][3];
                bands(t, sv, cv);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  // this lane's slot = 32 mt + drow(r, h): one of two compile-time slots; d slot / d x_ax = cc (zero for the padding slots)
                  float cc[2];
                  int axx[2];
#pragma unroll
                  for (int hh = 0; hh < 2; ++hh) {
                    const int slot = 32 * mt + drow(r, hh);
                    cc[hh] = 0.f;
                    axx[hh] = 0;
                    if (slot < 6 * L) {
                      const int pair = slot >> 1, f = pair / 3;
                      axx[hh] = pair - 3 * f;
                      const float wf_ = a.freq_w ? a.freq_w[f] : 1.0f;
                      // d/dx [w sin(2^f x)] = 2^f w cos ; d/dx [w cos(2^f x)] = -2^f w sin
                      cc[hh] = ldexpf((slot & 1) ? -sv[f < L ? f : 0][axx[hh]] : cv[f < L ? f : 0][axx[hh]], f) * wf_;
                    } else if (slot < 6 * L + 3) {
                      axx[hh] = slot - 6 * L;
                      cc[hh] = 1.f;
                    }
                  }
                  const float cg = (h ? cc[1] : cc[0]) * acc[t][r];
#pragma unroll
                  for (int k = 0; k < 3; ++k) {
                    const bool use = h ? (axx[1] == k) : (axx[0] == k);
                    dx[t][k] += use ? cg : 0.f;
                  }
                }
              }
            } else {
              // slot value > 0 <=> alive: the slot's pre-activation from the table (fp32, the same expression as the forward)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float4 row = ltab[32 * mt + drow(r, h)];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                  const float z = row.x * xs[t][0] + row.y * xs[t][1] + row.z * xs[t][2] + row.w;
                  const float dz = z > 0.f ? acc[t][r] : 0.f;
                  acc[t][r] = dz;
                  dx[t][0] += row.x * dz; dx[t][1] += row.y * dz; dx[t][2] += row.z * dz;
                }
              }
              // the masked slot gradients of this row tile -> units 2 mt, 2 mt + 1 of the dZ slot (layer 0's dZ there has been consumed)
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                uint4 u[2];
                tile_to_units<P>(acc[t], u);
                slabZ[(t * UW + 2 * mt) * 64 + lane] = u[0];
                slabZ[(t * UW + 2 * mt + 1) * 64 + lane] = u[1];
              }
            }
          }
          if constexpr (Net::EMB == 2) {
            if (a.g_aff != nullptr) {
              // [x_hi | 1 | x_lo | 0] as slots 0..7 of unit 0 (identity slot order: lane half 0), every other row of the 32-row tile zero
              __builtin_amdgcn_wave_barrier();
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                float hi[3], lo[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                  hi[k] = bf2f(f2bf(xs[t][k]));
                  lo[k] = xs[t][k] - hi[k];
                }
                const uint4 u0 = make_uint4(pack2bf(hi[0], hi[1]), pack2bf(hi[2], 1.0f), pack2bf(lo[0], lo[1]), pack2bf(lo[2], 0.f));
                slab0[xslot * SLOT + (t * UW + 0) * 64 + lane] = h ? make_uint4(0, 0, 0, 0) : u0;
                slab0[xslot * SLOT + (t * UW + 1) * 64 + lane] = make_uint4(0, 0, 0, 0);
              }
              __builtin_amdgcn_wave_barrier();
              const unsigned ldsP = lds0 + (unsigned)(xslot * SLOT * 16) + fo_emb;
              Frag fp[4], fe[4][2];
              sfor<0, 4>([&](auto kc) {
                constexpr int kk = decltype(kc)::value;
                frag_issue<0, kk>(ldsP, fp[kk]);
                frag_issue<0, kk>(ldsZ + fo_act, fe[kk][0]);
                frag_issue<1, kk>(ldsZ + fo_act, fe[kk][1]);
              });
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) frag_wait4(fp[kk], fe[kk][0], fe[kk][1], fp[kk]);
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) {
                mma_unit<P>(gA[0], frag_units(fe[kk][0]), frag_units(fp[kk]));
                mma_unit<P>(gA[1], frag_units(fe[kk][1]), frag_units(fp[kk]));
              }
              __builtin_amdgcn_wave_barrier();
            }
          }
        }
      }
    });
    if (a.d_x != nullptr) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int k = 0; k < 3; ++k) dx[t][k] += __shfl_xor(dx[t][k], 32, 64);
        if (h == 0 && sidx[t] < a.S) {
          a.d_x[(size_t)sidx[t] * 3 + 0] = dx[t][0];
          a.d_x[(size_t)sidx[t] * 3 + 1] = dx[t][1];
          a.d_x[(size_t)sidx[t] * 3 + 2] = dx[t][2];
        }
      }
    }
  }
  frame_flush();

  // ---- weight / bias gradients out: reduce the four waves' tiles in LDS (the slab space), one atomic per entry and workgroup ----
  __syncthreads();
  constexpr int NTILES = NH * 4 + 2;
  static_assert(NTILES * 1024 * 4 <= 4 * NSLOT * SLOT * 16, "reduction buffer");
  float* red = reinterpret_cast<float*>(slab_all);
  for (int e = threadIdx.x; e < NTILES * 1024; e += 256) red[e] = 0.f;
  __syncthreads();
  auto red_tile = [&](int ti, const f32x16_t& c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(red + ti * 1024 + drow(r, h) * 32 + n, c[r]);  // [row of the dZ tile][column of the X tile]
  };
#pragma unroll
  for (int l = 0; l < NH; ++l) {
    red_tile(l * 4 + 0, dWh[l][0][0]); red_tile(l * 4 + 1, dWh[l][0][1]); red_tile(l * 4 + 2, dWh[l][1][0]); red_tile(l * 4 + 3, dWh[l][1][1]);
  }
  red_tile(NH * 4 + 0, dWo[0]); red_tile(NH * 4 + 1, dWo[1]);
  __syncthreads();
  for (int e = threadIdx.x; e < NTILES * 1024; e += 256) {
    const int ti = e >> 10, rr = (e >> 5) & 31, cc = e & 31;
    const float v = red[e];
    if (v == 0.f) continue;
    if (ti < NH * 4) {
      const int l = ti >> 2, fa = (ti >> 1) & 1, fb = ti & 1;
      atomicAdd(a.dW[l] + (size_t)(32 * fa + rr) * 64 + 32 * fb + cc, v);
    } else {
      atomicAdd(a.dW[NH] + (size_t)rr * 64 + 32 * (ti - NH * 4) + cc, v);
    }
  }
  // bias gradients: column 0 of the row-sum tiles, one atomic per row and wave
#pragma unroll
  for (int l = 0; l < NH; ++l) {
    if (l == 0 && PF0) continue;  // left per frame by frame_flush
    if (a.db[l] == nullptr) continue;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (n == 0) atomicAdd(a.db[l] + 32 * mt + drow(r, h), dbh[l][mt][r]);
  }
  if (a.db[NH] != nullptr) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = dbo[r];
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off, 64);
      if (n == 0) atomicAdd(a.db[NH] + drow(r, h), v);
    }
  }
}

template <class Net>
int launch_mlp_bwd_fused(const FusedK& k, hipStream_t st);

#define LAB4D_MLP_INSTANTIATE_FUSED_BWD(Net)                                                                   \
  namespace lab4d {                                                                                            \
  template <>                                                                                                  \
  int launch_mlp_bwd_fused<Net>(const FusedK& k, hipStream_t st) {                                             \
    static int n_cu = 0;                                                                                       \
    if (n_cu == 0) {                                                                                           \
      int dev = 0;                                                                                             \
      hipDeviceProp_t prop;                                                                                    \
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256; \
      n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;                                    \
    }                                                                                                          \
    int g = (k.ntiles + 3) / 4;                                                                                \
    if (g > n_cu) g = n_cu;                                                                                    \
    hipLaunchKernelGGL((k_mlp_bwd_fused<Net>), dim3(g < 1 ? 1 : g), dim3(256), 0, st, k);                      \
    return check_launch("mlp_backward_fused");                                                                 \
  }                                                                                                            \
  }

}  // namespace lab4d
