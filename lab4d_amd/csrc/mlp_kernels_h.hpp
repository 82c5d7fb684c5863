// Backward (dgrad) chain for the 256-wide posenc nets with TWO waves per SIMD: 8-wave workgroups of 32-sample waves.
//
// The 4-wave kernel (mlp_kernels.hpp: k_mlp_bwd) gives a wave 64 samples, i.e. 128 registers of layer input, ~470 registers in all and
// one wave per SIMD: whenever that wave waits (LDS round trip of the weight groups, store issue, the step barrier) the SIMD's matrix
// pipe idles -- measured: 0.36 of the HBM roof, 29 % MFMA duty, and no single ablation recovers it.  Here a
// wave owns 32 samples (64 registers of layer input), eight waves share the same 2 x 14 KiB weight stream and the same 156 KiB of
// LDS, and the hardware overlaps one wave's epilogue / waits with the other's MFMAs.
// What makes 32-sample tiles practical is ds_read_b64_tr_b16 (tools/probes/tr_probe.hip): all tile IO between the blocked
// [64-sample block][feature][64] HBM layout and the accumulator layout goes through the wave's own slab slots with transposing
// reads -- 4 reads + 2 stores per 32 x 32 tile and no cross-lane VALU work:
//   store: the packed tile is in slab units (2mt, 2mt+1) anyway; group G = (h = G&1, a = G>>1), lane c = (m = c&3, j = c>>2) points at
//          sample 8m + j (+4 for the second read) of unit q; lane i then holds samples 8(i>>2)..+7 of feature 16q + 8a + 4h + (i&3).
//   load:  the tile's 128 16-byte row pieces are written lane-linear into the same slots (a [32 rows][64 B] image); group
//          G = (X = G&1, h = G>>1), lane c points at row 8a' + 4h + j, samples 16X + 4m..+3; lane i = sample 16X + i receives features
//          8a' + 4h + 0..3 as two packed dwords -- exactly accumulator registers 4a'..4a'+3 in packed order.
// The ReLU masks stay in the forward kernel's format ([64-tile][m-tile][lane], bits 8t + k + 16 odd for sample 2n + t): lane n of the
// half tile reads the word of forward lane 16 half + n/2 and uses the bits of t = n & 1.
// bf16 only; nets with a posenc input (EMB == 0).  Contract identical to k_mlp_bwd (include/lab4d_mlp.h).
#pragma once
#include "mlp_kernels.hpp"

namespace lab4d {

struct TrH { unsigned long long v[4]; };

__device__ __forceinline__ void trh_wait(TrH& r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3])); }
// store direction: v[2q + second]
__device__ __forceinline__ void trh_issue_store(unsigned addr, TrH& r) {
  asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
               "ds_read_b64_tr_b16 %1, %4 offset:64\n\t"
               "ds_read_b64_tr_b16 %2, %4 offset:1024\n\t"
               "ds_read_b64_tr_b16 %3, %4 offset:1088"
               : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3])
               : "v"(addr)
               : "memory");
}
// load direction: v[a'] = packed accumulator registers 4a'..4a'+3
__device__ __forceinline__ void trh_issue_load(unsigned addr, TrH& r) {
  asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
               "ds_read_b64_tr_b16 %1, %4 offset:512\n\t"
               "ds_read_b64_tr_b16 %2, %4 offset:1024\n\t"
               "ds_read_b64_tr_b16 %3, %4 offset:1536"
               : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3])
               : "v"(addr)
               : "memory");
}
__device__ __forceinline__ void trh_store(GLOBAL_AS void* buf, int F, int s0, int mt, int lane, const TrH& r) {
  const int i = lane & 15, G = lane >> 4, h = G & 1, a = G >> 1;
  GLOBAL_AS char* base = (GLOBAL_AS char*)buf + tile_base_offset<PBF16>(F, s0, 32 * mt);  // includes the (s0 & 63) column offset
  const unsigned lo = (unsigned)((8 * a + 4 * h + (i & 3)) * 128 + 16 * (i >> 2));
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const unsigned long long A = r.v[2 * q], B = r.v[2 * q + 1];
    gst16(base + (lo + (unsigned)(16 * q * 128)), (unsigned)A, (unsigned)(A >> 32), (unsigned)B, (unsigned)(B >> 32));
  }
}
// the two 16-byte row pieces lane L fetches of a [32 rows][32 samples] tile: pieces p = L and L + 64, row p >> 2, column p & 3
__device__ __forceinline__ void loadh_raw(const GLOBAL_AS void* buf, int F, int s0, int mt, int lane, uint4 (&raw)[2]) {
  const GLOBAL_AS char* base = (const GLOBAL_AS char*)buf + tile_base_offset<PBF16>(F, s0, 32 * mt);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int p = lane + 64 * k;
    raw[k] = gld16(base + (unsigned)((p >> 2) * 128 + (p & 3) * 16));
  }
}

template <class Net>
__global__ void __launch_bounds__(512) k_mlp_bwd_h(BwdK a) {
  static_assert(Net::EMB == 0, "posenc nets only");
  using P = PBF16;
  constexpr int NL = Net::NL, UW = Slab<Net, P>::UW, WAVES = 8, TILE = 32;
#ifdef LAB4D_H_ACG16
  constexpr int ACG = 16;  // experiment: every weight group through LDS (all 160 KiB used)
#else
  constexpr int ACG = ACACHE_G;
#endif
  constexpr int UNITS = UW * 64;  // uint4 slots per wave: one 32-sample n-tile
  __shared__ uint4 slab_all[WAVES * UNITS];
  __shared__ uint4 abuf[2 * ACG * 64];
  const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint4* wslab = slab_all + wid * UNITS;  // this wave's slab, slot (u, lane) at wslab[u * 64 + lane]
  uint4* slab = wslab + lane;
  const unsigned slab_lds = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)wslab;
  // per-lane supplier addresses of the transposing reads (mt = 0)
  unsigned trs_base, trl_base;
  {
    const int i = lane & 15, G = lane >> 4, m = i & 3, j = i >> 2;
    trs_base = slab_lds + (unsigned)((32 * (G & 1) + 8 * m + j) * 16 + 8 * (G >> 1));
    trl_base = slab_lds + (unsigned)((4 * (G >> 1) + j) * 64 + (16 * (G & 1) + 4 * m) * 2);
  }
  const int wave = blockIdx.x * WAVES + wid, nwaves = gridDim.x * WAVES;
  const int ntiles = a.S_pad / TILE;  // a multiple of 8 (S_pad % 256 == 0): the eight waves make the same number of trips

  for (int tile = wave; tile < ntiles; tile += nwaves) {
    const int s0 = tile * TILE;
    const int sidx = s0 + n;
    const bool live = sidx < a.S;
    // forward-format mask word of this lane: forward lane 32h + 16*(half tile) + n/2, bits of t = n & 1
    const int mlane = 32 * h + 16 * (tile & 1) + (n >> 1), mshift = 8 * (n & 1);
    float dx[3] = {0.f, 0.f, 0.f}, dx2[3] = {0.f, 0.f, 0.f};
    TrH trt;

    // ---- head gradient: (S, COUT) fp32 -> packed accumulator layout -> slab units 0,1 -> stored transposed ----
    {
      unsigned int w[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float v[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int f = drow(2 * k + o, h);
          v[o] = (f < Net::COUT && live) ? a.d_out[(size_t)sidx * Net::COUT + f] : 0.f;
        }
        w[k] = pack2bf(v[0], v[1]);
      }
      slab[0] = make_uint4(w[0], w[1], w[2], w[3]);
      slab[64] = make_uint4(w[4], w[5], w[6], w[7]);
      trh_issue_store(trs_base, trt);
      trh_wait(trt);
      trh_store((GLOBAL_AS void*)a.dz[NL - 1], pad32(Net::L[NL - 1].mout), s0, 0, lane, trt);
    }

#pragma nounroll
    for (int l = NL - 1; l >= 0; --l)
    sfor<0, NL>([&](auto ri) {
      constexpr int R = NL - 1 - decltype(ri)::value;
      if constexpr (bwd_rep<Net>(R) != R) return;
      constexpr unsigned MEMBERS = bwd_members<Net>(R);
      if (!((MEMBERS >> l) & 1u)) return;
      constexpr LS ls = Net::L[R];
      constexpr LS lp = Net::L[R > 0 ? R - 1 : 0];
      constexpr int GK = pad32(ls.mout) / P::FPG;
      constexpr int GL = GK < ACG ? GK : ACG, NQ = (GL + WAVES - 1) / WAVES;
      constexpr int MTE = ls.ke / 32, MTA = ls.kin / 32;
      constexpr int MTP = pad32(lp.mout) / 32;  // m-tiles of the layer below (mask row length)
      constexpr bool DO_ACT = (R > 0 && MTA > 0);
      const int lm1 = l > 0 ? l - 1 : 0;
      const GLOBAL_AS void* Wt = KARG_PTR(BwdK, const void*, WT, l);
      const GLOBAL_AS unsigned int* maskp = KARG_PTR(BwdK, const unsigned int*, mask, lm1);
      GLOBAL_AS void* dzp = KARG_PTR(BwdK, void*, dz, lm1);
      uint4 bin[GK];
#pragma unroll
      for (int u = 0; u < GK; ++u) bin[u] = slab[u * 64];

      auto mfma_tile = [&](auto has_pre, int pre, int rbuf, uint4 (&A)[GK], f32x16_t& acc) {
        constexpr bool PRE = decltype(has_pre)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < GK; ++g) {
          mma_unit<P>(acc, A[g], bin[g]);
          if constexpr (PRE) {
            if (g >= GL) A[g] = load_a(Wt, GK, pre, g, lane);
#ifndef LAB4D_ABL_HNOA  // kernel experiment (timing only, results wrong): no LDS reads of the shared weight groups
            else if (rbuf >= 0) A[g] = abuf[(rbuf * ACG + g) * 64 + lane];
#endif
          }
        }
      };
      auto a_fetch = [&](int mt, uint4 (&stg)[NQ]) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int g = wid + WAVES * i < GL ? wid + WAVES * i : GL - 1;
          stg[i] = load_a(Wt, GK, mt, g, lane);
        }
      };
      auto a_stash = [&](int buf, const uint4 (&stg)[NQ]) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int g = wid + WAVES * i < GL ? wid + WAVES * i : GL - 1;
          abuf[(buf * ACG + g) * 64 + lane] = stg[i];
        }
      };
      auto a_grab = [&](int buf, uint4 (&A)[GK]) {
#pragma unroll
        for (int g = 0; g < GL; ++g) A[g] = abuf[(buf * ACG + g) * 64 + lane];
      };
      // the same two-stage software pipeline over row tiles as k_mlp_bwd (progressive weight reload from the other LDS buffer)
      auto pipeline = [&](auto n_c, int tile0, auto&& pre, auto&& epi, auto&& fl) {
        constexpr int N = decltype(n_c)::value;
        if constexpr (N > 0) {
          uint4 A[GK], stg[NQ];
          f32x16_t acc0, acc1;
          wg_step_barrier();
          a_fetch(tile0, stg);
          a_stash(0, stg);
          if constexpr (N > 1) {
            a_fetch(tile0 + 1, stg);
            a_stash(1, stg);
          }
          a_fetch(tile0 + (N > 2 ? 2 : N - 1), stg);
#pragma unroll
          for (int g = GL; g < GK; ++g) A[g] = load_a(Wt, GK, tile0, g, lane);
          pre(0);
          wg_step_barrier();
          a_grab(0, A);
          constexpr int NSTEP = N - 1, NPAIR = NSTEP / 2;
          mfma_tile(std::bool_constant<(N > 1)>{}, tile0 + 1, 1, A, acc0);
          if constexpr (N > 1) {
            wg_step_barrier();
            a_stash(0, stg);
            a_fetch(tile0 + (N > 3 ? 3 : N - 1), stg);
          }
          if constexpr (NPAIR > 0) {
#pragma nounroll
            for (int k = 0; k < 2 * NPAIR; k += 2) {
              wg_step_barrier();
              mfma_tile(std::true_type{}, tile0 + (k + 2 < N ? k + 2 : N - 1), 0, A, acc1);
              epi(k, acc0);
              pre(k + 1);
              a_stash(1, stg);
              a_fetch(tile0 + (k + 4 < N ? k + 4 : N - 1), stg);
              fl(k);
              wg_step_barrier();
              mfma_tile(std::true_type{}, tile0 + (k + 3 < N ? k + 3 : N - 1), 1, A, acc0);
              epi(k + 1, acc1);
              pre(k + 2 < N ? k + 2 : N - 1);
              a_stash(0, stg);
              a_fetch(tile0 + (k + 5 < N ? k + 5 : N - 1), stg);
              fl(k + 1);
            }
          }
          if constexpr (NSTEP % 2 == 1) {
            wg_step_barrier();
            mfma_tile(std::false_type{}, 0, -1, A, acc1);
            epi(N - 2, acc0);
            fl(N - 2);
            pre(N - 1);
            epi(N - 1, acc1);
            fl(N - 1);
          } else {
            epi(N - 1, acc0);
            fl(N - 1);
          }
        }
      };
      uint4 raw[2];            // prefetched row pieces: stored embedding tile (epi_emb) or external gradient tile (epi_act)
      unsigned int mbits = 0;  // prefetched ReLU sign bits (forward format)
      // a [32 rows][32 samples] tile that `pre` fetched -> 8 packed dwords in accumulator order, through the slab slots of tile `u2`
      auto unpack_tile = [&](int u2, unsigned int (&we)[8]) {
        wslab[u2 * 128 + lane] = raw[0];       // image byte offset p * 16, p = lane (+ 64): slots (2 u2) * 64 + p
        wslab[u2 * 128 + 64 + lane] = raw[1];
        TrH t;
        trh_issue_load(trl_base + (unsigned)u2 * 2048u, t);
        trh_wait(t);
#pragma unroll
        for (int q = 0; q < 4; ++q) { we[2 * q] = (unsigned)t.v[q]; we[2 * q + 1] = (unsigned)(t.v[q] >> 32); }
      };
      auto pre_emb = [&](int mt) { loadh_raw((const GLOBAL_AS void*)a.emb, Net::KE, s0, mt, lane, raw); };
      auto pre_act = [&](int j) {
        if constexpr (lp.relu != 0) mbits = maskp[((size_t)(tile >> 1) * MTP + j) * 64 + mlane];
        if constexpr (lp.ext_grad != 0) loadh_raw((const GLOBAL_AS void*)a.ext_gin, pad32(lp.mout), s0, j, lane, raw);
      };
      auto no_flush = [&](int) {};
      // (a) gradient wrt the embedding slots -> input gradient
      auto epi_emb = [&](int mt, f32x16_t& acc) {
        unsigned int we[8];
        unpack_tile(mt, we);
        constexpr int L = Net::NFREQ;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int slot = 32 * mt + drow(r, h);
          const float gv = acc[r];
          if (slot < 6 * L) {
            const int pair = slot >> 1, f = pair / 3, ax = pair - 3 * f;
            // partner slot = register r ^ 1: the other half of the same packed dword
            const unsigned int pw_ = we[r >> 1];
            const float partner = bf2f((unsigned short)((r & 1) ? (pw_ & 0xffffu) : (pw_ >> 16)));
            const float c = ldexpf((r & 1) ? -partner : partner, f) * gv;
            dx[0] += ax == 0 ? c : 0.f;
            dx[1] += ax == 1 ? c : 0.f;
            dx[2] += ax == 2 ? c : 0.f;
          } else if (slot < 6 * L + 3) {
            const int ax = slot - 6 * L;
            dx[0] += ax == 0 ? gv : 0.f;
            dx[1] += ax == 1 ? gv : 0.f;
            dx[2] += ax == 2 ? gv : 0.f;
          } else if (Net::AUX3 && slot < 6 * L + 6) {
            const int ax = slot - 6 * L - 3;
            dx2[0] += ax == 0 ? gv : 0.f;
            dx2[1] += ax == 1 ? gv : 0.f;
            dx2[2] += ax == 2 ? gv : 0.f;
          }
        }
      };
      // (b) gradient wrt the previous layer's output -> masked dZ_{l-1}
      auto epi_act = [&](int j, f32x16_t& acc) {
        const unsigned int bits = mbits;
        if constexpr (lp.ext_grad != 0) {
          unsigned int we[8];
          unpack_tile(j, we);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            acc[2 * k] += bf2f((unsigned short)(we[k] & 0xffffu));
            acc[2 * k + 1] += bf2f((unsigned short)(we[k] >> 16));
          }
        }
        acc_fence(acc);
        unsigned int w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = pack2bf_op(acc[2 * k], acc[2 * k + 1]);
        if constexpr (lp.add_ext != 0) {
          // y = relu(z) + ext  ->  dL/dext = dL/dy: the unmasked tile, through the same slots
          slab[(2 * j) * 64] = make_uint4(w[0], w[1], w[2], w[3]);
          slab[(2 * j + 1) * 64] = make_uint4(w[4], w[5], w[6], w[7]);
          TrH t;
          trh_issue_store(trs_base + (unsigned)j * 2048u, t);
          trh_wait(t);
          trh_store((GLOBAL_AS void*)a.ext_gout, pad32(lp.mout), s0, j, lane, t);
        }
        {
          unsigned int alive = lp.relu != 0 ? (bits >> mshift) : 0xffffffffu;
          if (!live) alive = 0u;
#pragma unroll
          for (int k = 0; k < 8; ++k) w[k] = pk_mask_bf16(w[k], (alive >> k) & 0x00010001u);
        }
        slab[(2 * j) * 64] = make_uint4(w[0], w[1], w[2], w[3]);
        slab[(2 * j + 1) * 64] = make_uint4(w[4], w[5], w[6], w[7]);
        trh_issue_store(trs_base + (unsigned)j * 2048u, trt);
      };
      auto flush_act = [&](int j) {
        trh_wait(trt);
        trh_store(dzp, pad32(lp.mout), s0, j, lane, trt);
      };
      if constexpr (MTE > 0) {
        if (a.d_x != nullptr) pipeline(std::integral_constant<int, MTE>{}, 0, pre_emb, epi_emb, no_flush);
      }
      if constexpr (DO_ACT) pipeline(std::integral_constant<int, MTA>{}, MTE, pre_act, epi_act, flush_act);
    });

    if (a.d_x) {
#pragma unroll
      for (int k = 0; k < 3; ++k) dx[k] += __shfl_xor(dx[k], 32, 64);
      if (h == 0 && live) {
        a.d_x[(size_t)sidx * 3 + 0] = dx[0];
        a.d_x[(size_t)sidx * 3 + 1] = dx[1];
        a.d_x[(size_t)sidx * 3 + 2] = dx[2];
      }
      if constexpr (Net::AUX3) {
#pragma unroll
        for (int k = 0; k < 3; ++k) dx2[k] += __shfl_xor(dx2[k], 32, 64);
        if (h == 0 && live && a.d_x2) {
          a.d_x2[(size_t)sidx * 3 + 0] = dx2[0];
          a.d_x2[(size_t)sidx * 3 + 1] = dx2[1];
          a.d_x2[(size_t)sidx * 3 + 2] = dx2[2];
        }
      }
    }
  }
}

// 8-wave grid: one persistent workgroup per CU
inline int mlp_grid_h(int ntiles) {
  int g = (ntiles + 7) / 8;
  if (g > 256) g = 256;
  return g < 1 ? 1 : g;
}

}  // namespace lab4d
