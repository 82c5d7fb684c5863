// Weights-stationary chain kernels for the 256-wide posenc nets (bf16): forward chain k_mlp_fwd_ws, backward (dgrad) chain k_mlp_bwd_ws.
//
// The wave-resident kernels (mlp_kernels.hpp) give every wave 64 samples and stream each layer's 128 KiB of weights to every wave
// (L2 -> VGPR -> LDS -> VGPR, a workgroup barrier per 32-row step, one wave per SIMD at ~490 registers): 0.26 of the bf16 MFMA peak for four
// rounds, and round 4 measured that neither the schedule, nor the stores, nor an LDS-DMA weight stream is what bounds them.
// Here the dataflow is turned round (probe: tools/probes/ws_core.hip, 0.43-0.50 of the peak with stores and weight fetch):
//   * a 512-thread workgroup (8 waves, two per SIMD, <= 256 registers) owns 128 samples = two 64-sample blocks = four 32-sample n-tiles;
//   * wave w keeps ROW TILE w of the layer's weights (32 output features x K) in registers -- the same packed 1-KiB A groups the
//     wave-resident kernels read, fetched ONCE per layer and workgroup straight into VGPRs (no LDS staging, no per-step barrier);
//   * the layer input of all 128 samples lives in LDS as B units ([n-tile][unit][lane] x 16 B, the accumulator-layout units of
//     mlp_kernels.hpp) and is streamed through the MFMAs: one ds_read_b128 per v_mfma_f32_32x32x16_bf16;
//   * the wave converts / activates its 32 x 64 output slice and writes it as units (2w, 2w+1) of the OTHER half of a double-buffered
//     2 x 64 KiB activation slab: ONE workgroup barrier per layer (10 per 128 samples instead of 76 per 64);
//   * the next layer's A groups replace the current ones right behind their last MFMA (the fetch hides behind the last block's matrix work);
//   * tiles leave for HBM through ds_read_b64_tr_b16 (no VALU transposes), LATE and in PIECES: a layer's tile / sign-word stores are issued inside
//     the NEXT layer, a quarter tile every four k-groups between its MFMAs (WsSpread), behind that layer's own requests;
//   * shared bias rows live in LDS for the life of the workgroup, per-frame rows per tile (tiles inside one frame); the next layer's pointers are
//     scalar-loaded a layer ahead; the positional encoding is evaluated once per (sample, axis) into a scratch that aliases an activation buffer.
// Everything that reaches HBM has the layout and the VALUES of the wave-resident kernels (same packed weights, same accumulation order per
// output element: bias, then the k-groups in ascending order; same packed epilogue), so the two families are interchangeable launch by launch
// and are held bit-equal to each other in tests/test_gpu_mlp_ws.py.  bf16, EMB == 0 (posenc) nets whose widest layer is 256; training, inference
// and point-gradient-only modes.
// Measured (profiles/r04_ws_*.json, r04_clock_under_load.json): -25 % / -17 % shader cycles against the wave-resident forward /
// backward, -10 % time -- these kernels hold the package at its power cap and the denser one is clocked lower.
// Build switches: -DLAB4D_WS_TRACE (per-wave cycle trace, outputs wrong), -DLAB4D_WSABL_{NOFLUSH,NOST,NOTR,NOAPF,NOBIAS,NOPOSENC,HALFB} and -DLAB4D_ABL_L2STORE
// (timing-only ablations, results wrong), -DLAB4D_WS_LINEAR_STORE (lane-linear stores through ds_bpermute: correct, slower), -DLAB4D_WS_BD=n (B ring depth).
#pragma once
#include "mlp_kernels.hpp"

namespace lab4d {

template <class Net>
constexpr bool ws_ok() {
  if (!(Net::EMB == 0 && Net::AUX3 == 0 && net_wmax<Net>() == 256 && Net::KE % 32 == 0)) return false;
  for (int l = 0; l < Net::NL; ++l) {
    const int mt = pad32(Net::L[l].mout) / 32;
    if (!(mt == 1 || mt == 2 || mt == 4 || mt == 8)) return false;
    if (Net::L[l].kin % 32 != 0 || Net::L[l].kin > 256) return false;
    if (Net::L[l].ke != 0 && Net::L[l].ke != Net::KE) return false;
  }
  return true;
}
template <class Net>
constexpr int ws_g(int l) { return (Net::L[l].ke + Net::L[l].kin) / 16; }
template <class Net>
constexpr int ws_mt(int l) { return pad32(Net::L[l].mout) / 32; }
template <class Net>
constexpr int ws_gmax() {
  int g = 0;
  for (int l = 0; l < Net::NL; ++l) g = ws_g<Net>(l) > g ? ws_g<Net>(l) : g;
  return g;
}
// layers that share one copy of the layer code (runtime loop over the layers, see fwd_same): same shape, same number of A groups in the layer
// that FOLLOWS (its prefetch is unrolled into this layer's last MFMA loop), same shape of the layer in FRONT (whose stores this layer issues)
template <class Net>
constexpr bool wsf_same(int a, int b) {
  if (!fwd_same<Net>(a, b)) return false;
  const int an = (a + 1) % Net::NL, bn = (b + 1) % Net::NL;
  if (ws_g<Net>(an) != ws_g<Net>(bn)) return false;
  if ((a == 0) != (b == 0)) return false;
  if (a > 0 && ws_mt<Net>(a - 1) != ws_mt<Net>(b - 1)) return false;
  return true;
}
template <class Net>
constexpr int wsf_rep(int l) {
  for (int j = 0; j < l; ++j)
    if (wsf_same<Net>(j, l)) return j;
  return l;
}
template <class Net>
constexpr unsigned wsf_members(int r) {
  unsigned m = 0;
  for (int l = 0; l < Net::NL; ++l)
    if (wsf_rep<Net>(l) == r) m |= 1u << l;
  return m;
}
// row tiles of layer l as a runtime value (the layer loop is a runtime loop; l is wave-uniform)
template <class Net>
__device__ __forceinline__ int ws_mt_rt(int l) {
  int v = 1;
  sfor<0, Net::NL>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if (l == i) v = ws_mt<Net>(i);
  });
  return v;
}

constexpr int WS_TILE = 128;       // samples per workgroup tile
#ifndef LAB4D_WS_BD
#define LAB4D_WS_BD 2
#endif
constexpr int WS_BD = LAB4D_WS_BD;  // depth (k-groups) of the B-operand ring between LDS and the MFMAs (2 / 4 / 8 measured: the same time; 2 keeps the colour net free of spills)
constexpr int WS_BUF = 4 * 16 * 64;  // uint4 slots of one activation buffer: [n-tile 4][unit 16][lane 64] = 64 KiB

// this wave's work items of a layer with MT row tiles: an item = (row tile mt, 64-sample block b).  2 MT items over 8 waves:
// MT = 8: two items per wave (mt = w, b = 0, 1); MT = 4: one (mt = w & 3, b = w >> 2); fewer: waves >= 2 MT idle
template <int MT>
struct WsItems {
  static constexpr int ITEMS = 2 * MT, IPW = ITEMS >= 8 ? ITEMS / 8 : 1;
  __device__ __forceinline__ static bool active(int w) { return ITEMS >= 8 || w < ITEMS; }
  __device__ __forceinline__ static int mt(int w) { return w & (MT - 1); }
  __device__ __forceinline__ static int blk(int w, int k) { return MT >= 8 ? k : ((w / MT) & 1); }
};

// B-operand reads of the MFMA loops: volatile asm, so that they stay where they are written (ahead of their use by the depth of the ring)
template <int OFF>
__device__ __forceinline__ void ws_lds_read(u32x4_t& d, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// all but the N most recent LDS operations of this wave have returned (LDS returns in order; the operands tie the wait to its consumers)
template <int N>
__device__ __forceinline__ void ws_lds_wait(u32x4_t& a, u32x4_t& b) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt field");
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void ws_lds_wait4(u32x4_t (&r)[4]) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt field");
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N));
}
__device__ __forceinline__ void mma_b(f32x16_t& acc, const uint4& a, const u32x4_t& b) {
  bf16x8_t av;
  __builtin_memcpy(&av, &a, 16);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

// One quarter of a finished 32 x 64 tile on its way to HBM (rows 16 q + 8 a + .., qa = 2 q + a; 8 rows x 128 B = 1 KiB), in three stages placed
// BETWEEN the MFMAs of the layer that follows:
//   issue: two transposing LDS reads (see tr_issue): lane i of 16-lane group (h, S) then holds samples 32 S + 8 (i >> 2) .. + 7 of row 4 h + (i & 3);
//   perm (-DLAB4D_WS_LINEAR_STORE only): four ds_bpermute_b32 make the piece LANE-LINEAR (lane L holds bytes 16 L .. 16 L + 15 of the 1 KiB).
//          Straight out of the transposing reads the four lanes of a quad hold four different rows, and in isolation such a store costs the CU's
//          address path ~1.4x the cycles of a lane-linear one (tools/probes/store_pattern.hip: 47 vs 34 cycles per store beside MFMAs, 4.8 vs
//          5.5 TB/s); in the real kernels the four permutes cost 0.4 ms per 4.2 M samples and the linear stores save < 0.1 ms: off;
//   store: one global_store_dwordx4 per lane.
// Why spread at all: a CU writes 64 KiB per layer; issued as one burst at the top of the layer the stores block every wave's memory issue while
// the matrix pipe idles (measured: 6.4 ms per 4.2 M samples with the burst, 4.2 ms with the stores removed).
struct WsPiece {
  unsigned long long a, b;  // transposing reads
  unsigned int d[4];        // lane-linear dwords
};
// ---- slot swizzle of the activation slabs (round 6) ------------------------------------------------------------------------------------------
// A unit is 64 x 16 B, slot = lane = 32 h + n.  The transposing reads that take a finished tile out (ds_read_b64_tr_b16: two 32-lane groups, bank =
// (byte / 4) mod 64) address, per group, slots (t, h, n = 4 m + jj) for t, h in {0, 1}, m in 0..3, jj in {0, 1}: t is 16 KiB away, h 512 B -- both
// multiples of the 256-B bank row -- so FOUR lanes meet on every bank: 6 extra LDS cycles per read, 96 per wave and layer = the 21-25 % of
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE VERDICT r05 (weak #3) found in these kernels (profiles/r05_sq_counters_all.txt).  With the lanes of the
// upper half stored two slots over (slot = lane ^ 2 for h = 1: n bit 1, which the reads' own n never sets) the two h meet on different banks: 2-way
// (the t pair is left: both read the same 8-byte half of slots 16 KiB apart; separating them would need 8-byte interleaving, which the 16-byte B
// operands of the MFMAs cannot have).  The B reads (lane-linear ds_read_b128: 16-lane groups that never mix the two h) and the unit writes
// (ds_write_b128: aligned 8-lane groups) see a permutation inside their own groups: still conflict-free.  The second read of a pair (samples + 4 =
// slot n + 2) is `^ 32` on the byte address in both halves.  -DLAB4D_WS_SWZ=0 restores the plain layout (A/B measurements); results are
// bit-identical either way (tests/test_gpu_mlp_ws.py).
#ifndef LAB4D_WS_SWZ
#define LAB4D_WS_SWZ 1
#endif
constexpr bool WS_SWZ = LAB4D_WS_SWZ != 0;
__device__ __forceinline__ int ws_slot(int lane) { return WS_SWZ ? (lane ^ ((lane >> 5) << 1)) : lane; }
// per-lane part of the transposing tile reads (tr_lane_base of mlp_kernels.hpp with the swizzle): first read of a pair; the second is ^ 32
__device__ __forceinline__ unsigned ws_tr_lane(int lane) {
  const int i = lane & 15, G = lane >> 4, h = G & 1, S = G >> 1, m = i & 3, j = i >> 2;
  const int sigma = 32 * S + 8 * m + j, n = sigma >> 1, t = sigma & 1;
  return (unsigned)(((t * 16) * 64 + 32 * h + (WS_SWZ ? (n ^ (2 * h)) : n)) * 16);
}
// the whole-tile burst (tr_issue of mlp_kernels.hpp) with the pair's second address
__device__ __forceinline__ void ws_tr_issue(unsigned addr, unsigned addr2, TrTile& r) {
  asm volatile("ds_read_b64_tr_b16 %0, %8\n\t"
               "ds_read_b64_tr_b16 %4, %9\n\t"
               "ds_read_b64_tr_b16 %1, %8 offset:8\n\t"
               "ds_read_b64_tr_b16 %5, %9 offset:8\n\t"
               "ds_read_b64_tr_b16 %2, %8 offset:1024\n\t"
               "ds_read_b64_tr_b16 %6, %9 offset:1024\n\t"
               "ds_read_b64_tr_b16 %3, %8 offset:1032\n\t"
               "ds_read_b64_tr_b16 %7, %9 offset:1032"
               : "=&v"(r.a[0]), "=&v"(r.a[1]), "=&v"(r.a[2]), "=&v"(r.a[3]), "=&v"(r.b[0]), "=&v"(r.b[1]), "=&v"(r.b[2]), "=&v"(r.b[3])
               : "v"(addr), "v"(addr2)
               : "memory");
}
template <int QA>
__device__ __forceinline__ void ws_trp_issue(unsigned addr /* tile base + lane part */, unsigned addr2 /* tile base + (lane part ^ 32) */, WsPiece& r) {
  constexpr int OFF = 1024 * (QA >> 1) + 8 * (QA & 1);
#ifdef LAB4D_WSABL_NOTR  // timing experiment (results wrong): no transposing reads, the stores write whatever the registers hold
  r.a = addr; r.b = addr2 + OFF;
  return;
#endif
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\t"
               "ds_read_b64_tr_b16 %1, %3 offset:%4"
               : "=&v"(r.a), "=&v"(r.b)
               : "v"(addr), "v"(addr2), "n"(OFF));
}
// byte address (lane * 4) of the lane whose transposed piece lane L wants: L = 8 row + piece16  <-  group (h = row >> 2, S = piece16 >> 2), lane (row & 3) + 4 (piece16 & 3)
__device__ __forceinline__ unsigned ws_perm_addr(int lane) {
  const int row = lane >> 3, pc = lane & 7;
  return (unsigned)(4 * (16 * ((row >> 2) + 2 * (pc >> 2)) + (row & 3) + 4 * (pc & 3)));
}
// N = LDS operations this wave has certainly issued behind the stage's inputs (a lower bound keeps the wait safe)
template <int N>
__device__ __forceinline__ void ws_trp_perm(unsigned perm_addr, WsPiece& r) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt field");
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r.a), "+v"(r.b) : "n"(N));
  const unsigned a0 = (unsigned)r.a, a1 = (unsigned)(r.a >> 32), b0 = (unsigned)r.b, b1 = (unsigned)(r.b >> 32);
#ifndef LAB4D_WS_LINEAR_STORE
  r.d[0] = a0; r.d[1] = a1; r.d[2] = b0; r.d[3] = b1;  // the piece keeps the lane order of the transposing reads (ws_trp_store addresses it accordingly)
  return;
#endif
  asm volatile("ds_bpermute_b32 %0, %4, %5\n\t"
               "ds_bpermute_b32 %1, %4, %6\n\t"
               "ds_bpermute_b32 %2, %4, %7\n\t"
               "ds_bpermute_b32 %3, %4, %8"
               : "=&v"(r.d[0]), "=&v"(r.d[1]), "=&v"(r.d[2]), "=&v"(r.d[3])
               : "v"(perm_addr), "v"(a0), "v"(a1), "v"(b0), "v"(b1));
}
template <int QA, int N>
__device__ __forceinline__ void ws_trp_store(GLOBAL_AS void* buf, int F, int s0, int mt, int lane, WsPiece& r) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt field");
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r.d[0]), "+v"(r.d[1]), "+v"(r.d[2]), "+v"(r.d[3]) : "n"(N));
#ifdef LAB4D_WSABL_NOST  // timing experiment (results wrong): reads and waits, no store
  return;
#endif
#ifdef LAB4D_ABL_L2STORE  // timing experiment (results wrong): every wave keeps writing the same 2048-sample window, so the stores never reach HBM
  s0 &= 0x7ff;
#endif
  GLOBAL_AS char* base = (GLOBAL_AS char*)buf + tile_base_offset<PBF16>(F, s0, 32 * mt);
#ifdef LAB4D_WS_LINEAR_STORE
  const unsigned lo = (unsigned)(lane * 16);
#else
  const int i = lane & 15, G = lane >> 4, h = G & 1, S = G >> 1;
  const unsigned lo = (unsigned)((4 * h + (i & 3)) * 128 + (32 * S + 8 * (i >> 2)) * 2);
#endif
  gst16(base + (lo + (unsigned)((16 * (QA >> 1) + 8 * (QA & 1)) * 128)), r.d[0], r.d[1], r.d[2], r.d[3]);
}
// Where the NPI pieces of one hosted tile set go in a loop of G k-groups with a B ring of depth BD: piece i is read at group I = (i G) / NPI, made
// lane-linear at I + 1 and stored at I + 3 (one piece in flight: G / NPI >= 4).
template <int G, int NPI>
struct WsSpread {
  static constexpr bool OK = NPI > 0 && G >= 4 * NPI;
  static constexpr int issue_at(int i) { return (i * G) / NPI; }
  static constexpr int perm_at(int i) { return issue_at(i) + 1; }
  static constexpr int store_at(int i) { return issue_at(i) + 3; }
  // LDS operations certainly issued between a stage placed in group g0 (in front of that group's B read-ahead) / g0 (behind it) and a stage behind the
  // read-ahead of group g1: the B reads (two per group) of the groups in between that still read ahead
  static constexpr int newer(int g0, bool g0_incl, int g1, int BD) {
    int c = 0;
    for (int j = g0_incl ? g0 : g0 + 1; j <= g1; ++j) c += (j + BD < G) ? 2 : 0;
    return c;
  }
};

// -DLAB4D_WS_TRACE (measurement build, outputs wrong): the forward kernel sums, per wave, the shader cycles it spends in the phases of a layer
// and waves 0 and 4 of workgroup 0 leave the sums in the first floats of `out` (tools/ws_compare.py --trace prints them)
__device__ __forceinline__ unsigned long long ws_clock() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#ifdef LAB4D_WS_TRACE
#define WS_T(i)                                   \
  do {                                            \
    const unsigned long long t_ = ws_clock();     \
    tacc[i] += (float)(unsigned)(t_ - tlast);     \
    tlast = t_;                                   \
  } while (0)
#else
#define WS_T(i) \
  do {          \
  } while (0)
#endif

// End-of-layer barrier that also settles the scalar loads of the NEXT layer's pointers: the wait is the BUILTIN (the compiler's counter model
// sees it: nothing scalar is pending afterwards), so the first LDS waits of the next layer are the hand-counted ones of the B ring and the first MFMA
// waits for its own two reads only -- with a pointer load in flight the compiler has to wait lgkmcnt(0) (scalar loads return out of order), i.e. for
// the whole ring prologue of all eight waves (trace build: 500-900 cycles per layer between the barrier and the first MFMA).
__device__ __forceinline__ void ws_layer_barrier() {
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), vmcnt / expcnt untouched
  asm volatile("s_barrier" ::: "memory");
}
// ---- per-block progress counters (LAB4D_WS_SYNC = 1, round 5) -------------------------------------------------------------------------------
// The per-layer s_barrier keeps all eight waves in lock-step: the two waves of a SIMD enter their MFMA loops together and leave them together, so the
// matrix pipe idles while both run their epilogues (profiles/r04_ws_trace.json: the older wave of every SIMD parks 22-28 k of ~88 k cycles per tile at
// the barrier).  What a layer really needs is per 64-sample BLOCK: item (layer l, block b) reads block b of layer l-1's output (all eight row tiles) and
// overwrites block b of the buffer layer l-1 read -- both settled once every wave has finished ITS item (l-1, b).  So each wave counts itself into
// cnt[b] when it is done with block b of a layer (its LDS writes are performed: the LDS pipe is in-order per wave and the wait in front of the add makes
// it explicit) and polls cnt[b] before it touches block b of the next layer.  A wave that finishes early runs up to one item ahead of the slowest one;
// the two waves of a SIMD drift out of phase and one's epilogue runs under the other's MFMAs.  Counters are monotone over the life of the workgroup
// (8 arrivals per block and layer); the tile boundary keeps its full barriers (posenc scratch / embedding / head-gradient staging alias the buffers).
// MEASURED (round 5, profiles/r05_ws_sync_token.json; ms per 4.2 M samples, fg base forward / backward, fg colour forward / backward):
//   barrier (shipped)        5.996 / 6.796   3.142 / 3.100
//   counters                 5.988 / 6.937   3.400 / 3.415
//   counters + token         6.003 / 7.014   3.137 / 3.586
//   barrier + token          6.427 / 7.278   3.212 / 3.273
// i.e. NOT a win: the older wave of a SIMD no longer parks at the barrier, it parks in ws_wait_block instead (14-17 k of 86 k cycles per tile) -- the
// critical path is the YOUNGER wave of each SIMD, which is busy the whole tile (its two loops run at half rate while the older wave's run beside them), and
// one item of slack does not move work from it.  Bit-equal to the barrier build on hardware in every mode (tests/test_gpu_mlp_ws.py ran with
// counters + token).  Kept as an experiment switch, default OFF.
#ifndef LAB4D_WS_SYNC
#define LAB4D_WS_SYNC 0
#endif
constexpr bool WS_SYNC = LAB4D_WS_SYNC != 0;
__device__ __forceinline__ void ws_arrive(unsigned cnt_addr, int lane) {
  // (memory clobber: the epilogue's LDS stores are emitted in front of this; lgkmcnt(0): they have been performed)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) {
    const unsigned one = 1u;
    asm volatile("ds_add_u32 %0, %1" ::"v"(cnt_addr), "v"(one) : "memory");
  }
}
__device__ __forceinline__ void ws_wait_block(unsigned cnt_addr, unsigned target) {
  unsigned v;
  int spins = 0;
  for (;;) {
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cnt_addr) : "memory");
    if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)v) - target) >= 0) break;
    if (++spins > (1 << 22)) __builtin_trap();  // a lost arrival must fail loudly, not hang the device
    __builtin_amdgcn_s_sleep(1);
  }
}

// ---- matrix-pipe token (LAB4D_WS_TOKEN = 1, round 5) ------------------------------------------------------------------------------------------
// Two waves share a SIMD (wave w and w + 4).  Left alone they enter their MFMA loops together (each at half rate), leave them together and run their
// epilogues together with the matrix pipe idle: profiles/r05_ws_trace_*.json -- the SIMD's pipe is busy 41 k of 85 k cycles per tile.  What
// the dataflow wants is ALTERNATION: one wave streams its item through the pipe at full rate while the other converts / stores / waits.  The token is a
// per-SIMD lock in LDS taken in front of an item's MFMA loop (behind the block wait) and dropped behind its last MFMA; nothing is waited for while it is
// held (no deadlock), and the counters above let the other wave run its epilogue and the next item's entry in the meantime.
// MEASURED (table above): slower.  A lone wave streams an item in ~1.45 k cycles (1,024 of them MFMA: the depth-2 B ring does not cover the LDS latency
// without a second wave's instructions in between), so serialising the two waves' loops costs more than their overlapping epilogues return
// (95 k instead of 85 k cycles per tile).  Experiment switch, default OFF.
#ifndef LAB4D_WS_TOKEN
#define LAB4D_WS_TOKEN 0
#endif
constexpr bool WS_TOKEN = LAB4D_WS_TOKEN != 0;
__device__ __forceinline__ void ws_token_acquire(unsigned tok_addr, int lane) {
  if constexpr (!WS_TOKEN) return;
  const unsigned one = 1u;
  int spins = 0;
  for (;;) {
    unsigned old = 1u;
    if (lane == 0) asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(tok_addr), "v"(one) : "memory");
    if (__builtin_amdgcn_readfirstlane((int)old) == 0) break;  // (lane 0 is the first active lane)
    if (++spins > (1 << 22)) __builtin_trap();
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ void ws_token_release(unsigned tok_addr, int lane) {
  if constexpr (!WS_TOKEN) return;
  const unsigned zero = 0u;
  if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(tok_addr), "v"(zero) : "memory");
}

template <class T>
__device__ __forceinline__ void ws_pin_sgpr(T*& p) {
  unsigned long long v = (unsigned long long)p;
  asm volatile("" : "+s"(v));
  p = (T*)v;
}

// compile-time checks of the piece schedule for every (G, NPI) the kernels instantiate: one piece in flight (a piece is stored before the next one is
// read), every stage inside its loop, every counted wait encodable and never above what was really issued in between
template <int G, int NPI>
constexpr bool ws_spread_ok() {
  using SP = WsSpread<G, NPI>;
  for (int i = 0; i < NPI; ++i) {
    if (!(SP::issue_at(i) < SP::perm_at(i) && SP::perm_at(i) < SP::store_at(i) && SP::store_at(i) < G)) return false;
    if (i + 1 < NPI && !(SP::store_at(i) < SP::issue_at(i + 1))) return false;
    for (int bd = 1; bd <= 8; ++bd) {
      const int n0 = SP::newer(SP::issue_at(i), true, SP::perm_at(i), bd), n1 = SP::newer(SP::perm_at(i), false, SP::store_at(i), bd);
      if (n0 < 0 || n0 > 4 || n1 < 0 || n1 > 4) return false;  // two groups at two reads each, at most
    }
  }
  return true;
}
static_assert(ws_spread_ok<16, 4>() && ws_spread_ok<20, 4>() && ws_spread_ok<8, 2>() && ws_spread_ok<16, 2>(), "WsSpread schedule");
static_assert(!WsSpread<16, 8>::OK && !WsSpread<4, 4>::OK, "layers too short to host their pieces fall back to the burst");

// ---- LDS byte address of the uint4 array element (address space 3 pointers are 32 bit) ----
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p; }

// =================================================================================================
// forward chain, weights stationary
// =================================================================================================
// ST = training mode: embedding, every hidden post-activation and every ReLU sign word are stored (pointers host-checked);
// !ST = inference: only a layer another net consumes (act[l] != NULL) is stored.
// ACTS = false (with ST): the point-gradient-only mode of the sdf basefields (eval normals, nerf.py:455-493): embedding and sign words only
// TAN = tangent mode (the eikonal term, see k_mlp_fwd): the input is a raw (S, KE) tangent vector in embedding-slot order, no biases, ReLU replaced by the
// sign words the primal pass stored (read, not written); the tangent activations and the tangent embedding are stored for the weight gradients
template <class Net, bool ST, bool ACTS = true, bool TAN = false>
__global__ void __launch_bounds__(512) k_mlp_fwd_ws(FwdK a) {
  static_assert(ws_ok<Net>(), "weights-stationary chain: 256-wide posenc nets only");
  using P = PBF16;
  constexpr int NL = Net::NL, KE = Net::KE, UE = KE / 16, GMAX = ws_gmax<Net>(), L = Net::NFREQ;
  constexpr int ESTR = KE + 4;  // fp32 row stride of the posenc scratch (conflict-free 16-byte row reads for KE = 64 and 96)
  static_assert(WS_TILE * ESTR * 4 <= WS_BUF * 16, "posenc scratch aliases one activation buffer");
  __shared__ uint4 xbuf[2 * WS_BUF];
  __shared__ uint4 ebuf[4 * UE * 64];  // embedding as B units: [n-tile][unit][lane]
  // every layer's bias row (shared bias: filled once per workgroup; per-frame bias: the current tile's frame, refilled per tile when the tile lies in one
  // frame).  Read from here a bias costs LDS reads, not vector-memory loads that queue -- in the in-order memory counter -- behind the tile stores
  // of the layer in front and hold up the first MFMA of every layer.
  __shared__ float bias_lds[NL * 256];
  __shared__ unsigned int blk_cnt[2 + 4];  // [0..1] per 64-sample block: waves that have finished it, summed over the layers (see ws_arrive); [2..5] per SIMD: the matrix-pipe token
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned xbuf_lds = lds_addr(xbuf), ebuf_lds = lds_addr(ebuf), cnt_lds = lds_addr(blk_cnt), tok_lds = cnt_lds + 4u * (2u + (unsigned)(w & 3));
  if constexpr (WS_SYNC || WS_TOKEN) {  // (the shipped barrier-only build neither initialises nor touches the counters / tokens: ADVICE r05)
    if (tid < 6) blk_cnt[tid] = 0u;  // (the tile loop's first barrier is in front of every use)
  }
  unsigned cnt_base = 0u;          // arrivals per block before the current tile
  sfor<0, NL>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    if constexpr (TAN) {
      if (tid < 256) bias_lds[l * 256 + tid] = 0.f;  // tangent mode: no bias anywhere
    } else if constexpr (Net::L[l].pf == 0) {
      if (tid < 32 * ws_mt<Net>(l)) bias_lds[l * 256 + tid] = ((const GLOBAL_AS float*)a.bias[l])[tid];
    }
  });
  const unsigned trl = ws_tr_lane(lane), trl2 = trl ^ 32u;  // per-lane part of the transposing tile reads (unit stride 16 per n-tile): first / second read of a pair
  const int lane_s = ws_slot(lane);                          // this lane's slot inside a unit of the activation slabs
  const unsigned perm_a = ws_perm_addr(lane);

  int S_eff = a.S, ntw = a.S_pad / WS_TILE;
  if (a.S_dev) {
    const int sd = *(const GLOBAL_AS int*)a.S_dev;
    S_eff = sd < a.S ? sd : a.S;
    const int nt = (S_eff + WS_TILE - 1) / WS_TILE;
    ntw = nt < ntw ? nt : ntw;
  }

  // Round 5: what a tile needs from HBM before its first barrier is asked for a TILE AHEAD -- the point coordinates of the positional encoding (one
  // float per thread: sample pe_sl, axis pe_q) -- and the per-frame bias rows are refilled only when the tile's frame differs from the previous tile's (a
  // frame is 65,536 tiles long in the bench).  Before, every tile opened with a dependent HBM round trip that nothing could hide (all eight waves wait
  // for it at the tile's first barrier): -2 .. -4 % on the chains (profiles/r05_ws_prefetch.json).
  // HIDE: the nets' heads have ONE row tile (sdf: 1 output, rgb / dense: 3), so in a tile's last layer only waves 0 and 1 have an item; waves 2 .. 7 --
  // 384 threads = 128 samples x 3 axes -- evaluate the NEXT tile's positional encoding meanwhile (sincos, octave doubling, fp32 scratch rows), into the
  // activation buffer the last layer does not read.  The tile then opens with the scratch -> B-unit conversion straight away: one barrier and the whole
  // sincos phase (7-15 % of a tile, profiles/r05_ws_trace_call1.txt "posenc") leave the critical path.
  // MEASURED (round 5, two boxes, profiles/r05_ws_prefetch.json): not a win -- fg base forward 5.89 -> 5.98 ms per 4.2 M samples, fg colour forward 3.01 -> 3.16
  // (the six waves' sincos work competes for issue slots with the head item of waves 0 / 1, which is the layer's critical path, and the colour kernel
  // spills 17 registers around it).  Experiment switch LAB4D_WS_HIDE_POSENC, default OFF; the tile-ahead fetch above stays.
#ifndef LAB4D_WS_HIDE_POSENC
#define LAB4D_WS_HIDE_POSENC 0
#endif
  constexpr bool HIDE = LAB4D_WS_HIDE_POSENC != 0 && !TAN && ws_mt<Net>(NL - 1) == 1 && NL >= 2;
  constexpr int SCR_BUF = HIDE ? (((NL - 1) & 1) ^ 1) : 1;  // HIDE: the buffer the last layer's input is NOT in (its own output goes to HBM)
  float* scr = reinterpret_cast<float*>(xbuf + SCR_BUF * WS_BUF);
  // (the thread -> (sample, axis) map is recomputed where it is used: three registers less across the MFMA loops)
#define WS_PE_MAP                                                                             \
  const int pe_t = HIDE ? tid - 128 : tid;                                                    \
  const bool pe_on = pe_t >= 0 && pe_t < 384;                                                 \
  const int pe_sl = pe_t & 127, pe_q = __builtin_amdgcn_readfirstlane((pe_t >> 7) & 3)
  auto x_fetch = [&](int tile_, float& xp) {
    if constexpr (!TAN) {
      WS_PE_MAP;
      const int s = tile_ * WS_TILE + pe_sl, sc = s < S_eff ? s : S_eff - 1;
      if (tile_ < ntw && pe_on) xp = a.x[(size_t)sc * 3 + pe_q];
    }
  };
  // phase 1 of a tile's positional encoding: thread (sample pe_sl, axis pe_q) -> its columns of the fp32 scratch row.  Same arithmetic as k_mlp_fwd's
  // bf16 path: one accurate sincos per axis, angle doubling per octave, times the annealing weight; the raw coordinate; axis 0 also clears the padding.
  auto posenc_rows = [&](float xa) {
    if constexpr (!TAN) {
      WS_PE_MAP;
      if (pe_on) {
        const int row = 64 * (pe_sl >> 6) + 32 * (pe_sl & 1) + ((pe_sl & 63) >> 1);  // rows ordered (block, n-tile, lane n)
        float* dst = scr + row * ESTR;
        float sn, cs;
        sincosf(xa, &sn, &cs);
#pragma unroll
        for (int f = 0; f < L; ++f) {
          const float wf = a.freq_w ? a.freq_w[f] : 1.0f;
          *reinterpret_cast<float2*>(dst + 6 * f + 2 * pe_q) = make_float2(sn * wf, cs * wf);
          sincos_double(sn, cs);
        }
        dst[6 * L + pe_q] = xa;
        if (pe_q == 0) {
#pragma unroll
          for (int c = 6 * L + 3; c < KE; ++c) dst[c] = 0.f;
        }
      }
    }
  };
#undef WS_PE_MAP
  float xcur = 0.f;
  x_fetch((int)blockIdx.x, xcur);
  if constexpr (HIDE) {
#ifndef LAB4D_WSABL_NOPOSENC
    if ((int)blockIdx.x < ntw) posenc_rows(xcur);  // the first tile's; every later tile's is evaluated inside the previous tile's last layer
#endif
    wg_step_barrier();
  }
  uint4 A[GMAX];  // this wave's row tile of the current layer's weights
  auto a_load = [&](auto g0c, auto g1c, const GLOBAL_AS void* Wp, int G, int mt) {
    constexpr int G0 = decltype(g0c)::value, G1 = decltype(g1c)::value;
#pragma unroll
    for (int g = G0; g < G1; ++g) A[g] = load_a(Wp, G, mt, g, lane);
  };
  if ((int)blockIdx.x < ntw) {
    constexpr int G0 = ws_g<Net>(0), MT0 = ws_mt<Net>(0);
    a_load(std::integral_constant<int, 0>{}, std::integral_constant<int, G0>{}, KARG_PTR(FwdK, const void*, W, 0), G0, w & (MT0 - 1));
  }
  unsigned int pbits[2] = {0u, 0u};  // ReLU sign words of the layer just finished, waiting for their (deferred) store
  TrTile trt;
  // pointers the NEXT layer needs, loaded (scalar) in front of this layer's closing barrier: the following layer's weights, this layer's activation /
  // sign-word buffers (the next layer issues their stores)
  const GLOBAL_AS void* Wn_c = KARG_PTR(FwdK, const void*, W, (NL > 1 ? 1 : 0));
  GLOBAL_AS void* act_c = nullptr;
  GLOBAL_AS unsigned int* mask_c = nullptr;
#ifdef LAB4D_WS_TRACE
  float tacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  unsigned long long tlast = ws_clock();
#endif

  int f_bias = -1;  // frame whose per-frame bias rows bias_lds holds
  for (int tile = blockIdx.x; tile < ntw; tile += gridDim.x) {
    const int s0 = tile * WS_TILE;
    float xnext = 0.f;
    x_fetch(tile + (int)gridDim.x, xnext);
    // frame of this lane's sample in (block b, n-tile t): sample s0 + 64 b + 2 n + t
    int frame[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int s = s0 + 64 * b + 2 * n + t;
        const int sc = s < S_eff ? s : S_eff - 1;  // padded tail recomputes the last sample (finite, never written out)
        frame[b][t] = a.frame_idx ? ((const GLOBAL_AS int*)a.frame_idx)[sc] : sc / a.spf;
      }

    // per-frame bias rows of this tile's frame -> LDS (tiles inside one frame: the training shapes); tiles that straddle frames (and the compacted
    // evaluation, whose samples name their frames one by one) read theirs per lane from global memory
    bool tile_uni = TAN;
    if (!TAN && a.frame_idx == nullptr) {
      const int sl_ = s0 + WS_TILE - 1 < S_eff ? s0 + WS_TILE - 1 : S_eff - 1, sf_ = s0 < S_eff ? s0 : S_eff - 1;
      const int f0 = sf_ / a.spf, f1 = sl_ / a.spf;
      tile_uni = f0 == f1;
      if (tile_uni && f0 != f_bias) {
        f_bias = f0;
        sfor<0, NL>([&](auto lc) {
          constexpr int l = decltype(lc)::value;
          if constexpr (Net::L[l].pf != 0) {
            constexpr int MO = 32 * ws_mt<Net>(l);
            if (tid < MO) bias_lds[l * 256 + tid] = ((const GLOBAL_AS float*)a.pf_bias[l])[(size_t)f0 * MO + tid];
          }
        });
      }
    }
    // ---- positional encoding, phase 1 (see posenc_rows); HIDE: already done inside the previous tile's last layer ----
#ifndef LAB4D_WSABL_NOPOSENC
    if constexpr (TAN) {
      // raw (S, KE) tangent rows -> the scratch rows the assembly below reads (16-byte pieces, coalesced; elements behind the last valid one repeat it,
      // like stage_in of the wave-resident kernel)
      const long e_last = (long)S_eff * KE - 1;
      for (int p = tid; p < WS_TILE * (KE / 4); p += 512) {
        const int sl = p / (KE / 4), c4 = p - sl * (KE / 4);
        const int row = 64 * (sl >> 6) + 32 * (sl & 1) + ((sl & 63) >> 1);
        const long e0 = (long)(s0 + sl) * KE + 4 * c4;
        float4 v;
        if (e0 + 3 <= e_last) {
          const f32x4_t g4 = *(const GLOBAL_AS f32x4_t*)((const GLOBAL_AS float*)a.x + e0);
          v = make_float4(g4.x, g4.y, g4.z, g4.w);
        } else {
          const GLOBAL_AS float* gx = (const GLOBAL_AS float*)a.x;
          v = make_float4(gx[e0 <= e_last ? e0 : e_last], gx[e0 + 1 <= e_last ? e0 + 1 : e_last], gx[e0 + 2 <= e_last ? e0 + 2 : e_last], gx[e0 + 3 <= e_last ? e0 + 3 : e_last]);
        }
        *reinterpret_cast<float4*>(scr + row * ESTR + 4 * c4) = v;
      }
    } else if constexpr (!HIDE) {
      posenc_rows(xcur);
    }
#endif
    if constexpr (!HIDE) wg_step_barrier();
    // ---- scratch -> B units (identity slot order: unit g of lane (n, h) = slots 16 g + 8 h + 0..7) + the stored [slot][sample] embedding ----
    for (int p = w; p < 2 * UE; p += 8) {
      const int b = p & 1, g = p >> 1;
      uint4 emb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float4* r = reinterpret_cast<const float4*>(scr + (64 * b + 32 * t + n) * ESTR + 16 * g + 8 * h);
        const float4 v0 = r[0], v1 = r[1];
        emb[t] = make_uint4(pack2bf(v0.x, v0.y), pack2bf(v0.z, v0.w), pack2bf(v1.x, v1.y), pack2bf(v1.z, v1.w));
        ebuf[((2 * b + t) * UE + g) * 64 + lane] = emb[t];
      }
      if constexpr (ST) {
        const int q = n & 3, kq = n >> 2;
        GLOBAL_AS char* base = (GLOBAL_AS char*)a.emb + tile_base_offset<P>(KE, s0 + 64 * b, 0);
        const unsigned int w0[4] = {emb[0].x, emb[0].y, emb[0].z, emb[0].w};
        const unsigned int w1[4] = {emb[1].x, emb[1].y, emb[1].z, emb[1].w};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          unsigned int d[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int j = 4 * half + jj;
            const unsigned int lo = (w0[j >> 1] >> (16 * (j & 1))) & 0xffffu, hi = (w1[j >> 1] >> (16 * (j & 1))) & 0xffffu;
            d[jj] = lo | (hi << 16);
          }
          quad_transpose(d, q);
          gst16(base + tile_lane_offset<P>(16 * g + 8 * h + 4 * half + q, kq), d[0], d[1], d[2], d[3]);
        }
      }
    }
    wg_step_barrier();
    WS_T(6);

    // ---- layers ----
#pragma nounroll
    for (int l = 0; l < NL; ++l)
    sfor<0, NL>([&](auto ri) {
      constexpr int R = decltype(ri)::value;
      if constexpr (wsf_rep<Net>(R) != R) return;
      constexpr unsigned MEMBERS = wsf_members<Net>(R);
      if (!((MEMBERS >> l) & 1u)) return;
      constexpr LS ls = Net::L[R];
      constexpr bool LAST = (R == NL - 1);
      constexpr int MT = ws_mt<Net>(R), GE = ls.ke / 16, GA = ls.kin / 16, G = GE + GA;
      constexpr int Gn = ws_g<Net>((R + 1) % NL);  // A groups of the layer that follows (layer 0 of the next tile behind the last)
#ifdef LAB4D_WS_TRACE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trace build: how long the layer's first instructions would wait for the weights / the stores in front of them
      WS_T(7);
#endif
      using IT = WsItems<MT>;
      const GLOBAL_AS float* bl = KARG_PTR(FwdK, const float*, bias, l);
      const GLOBAL_AS float* pfl = KARG_PTR(FwdK, const float*, pf_bias, l);
      const int ln = l + 1 < NL ? l + 1 : 0;
      const GLOBAL_AS void* Wn = Wn_c;
      const int mtn = w & (ws_mt_rt<Net>(ln) - 1);
      const int ib = l & 1;
      const uint4* xin = xbuf + ib * WS_BUF;
      uint4* xout = xbuf + (ib ^ 1) * WS_BUF;
      const int mt = IT::mt(w);
      const bool active = IT::active(w);
      const unsigned cnt_tgt = cnt_base + 8u * (unsigned)l;  // every wave has finished layer l - 1 on a block once its counter reads this

      // bias (+ per-frame bias, which already contains the shared one: host contract) of item (mt, b) in accumulator layout
      // bias row from LDS (shared bias, or a tile inside one frame): four reads issued here, waited for (counted) behind the first item's ring prologue
      const bool bias_lds_path = ls.pf == 0 || tile_uni;
      u32x4_t br[4];
      auto load_bias = [&](int b, f32x16_t (&bv)[2]) {
        if (bias_lds_path) return;  // (same row for every item)
#pragma unroll
        for (int t = 0; t < (ls.pf != 0 ? 2 : 1); ++t) {
          const int fr = frame[0][t] + b * (frame[1][t] - frame[0][t]);  // b is 0 / 1 (arithmetic, not an index: an indexed private array goes to LDS / scratch)
          const GLOBAL_AS float* src = (ls.pf != 0) ? pfl + (size_t)fr * (32 * MT) : bl;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4_t v = *(const GLOBAL_AS f32x4_t*)(src + 32 * mt + 8 * i + 4 * h);
            bv[t][4 * i + 0] = v.x; bv[t][4 * i + 1] = v.y; bv[t][4 * i + 2] = v.z; bv[t][4 * i + 3] = v.w;
          }
        }
        if constexpr (ls.pf == 0) bv[1] = bv[0];
      };
      uint4 ext_raw[4];
      auto load_ext = [&](int b) {
        if constexpr (ls.add_ext != 0) load_tile_raw<P>((const GLOBAL_AS void*)a.ext, 32 * MT, s0 + 64 * b, mt, lane, ext_raw);
      };

      // ---- requests go out BEFORE the stores of the layer in front: the next layer's A groups this layer has no use for, the first item's bias / ext tile ----
#pragma unroll
      for (int g = G; g < Gn; ++g) A[g] = load_a(Wn, Gn, mtn, g, lane);
      unsigned int tbits[2] = {0xffffffffu, 0xffffffffu};  // tangent mode: the primal's sign words of this wave's items
      if constexpr (TAN && ls.relu != 0 && !LAST) {
        if (active) {
          const GLOBAL_AS unsigned int* mown = KARG_PTR(FwdK, unsigned int*, mask, l);
#pragma unroll
          for (int k = 0; k < IT::IPW; ++k) tbits[k] = mown[((size_t)(2 * tile + IT::blk(w, k)) * MT + mt) * 64 + lane];
        }
      }
      f32x16_t bv[2];
#ifdef LAB4D_WSABL_NOBIAS
#pragma unroll
      for (int r = 0; r < 16; ++r) bv[0][r] = bv[1][r] = 0.25f;
#else
      if (active) {
        if (bias_lds_path) {
          const unsigned ba = lds_addr(bias_lds) + (unsigned)((l * 256 + 32 * mt + 4 * h) * 4);
          ws_lds_read<0>(br[0], ba);
          ws_lds_read<32>(br[1], ba);
          ws_lds_read<64>(br[2], ba);
          ws_lds_read<96>(br[3], ba);
        } else {
          load_bias(IT::blk(w, 0), bv);
        }
        load_ext(IT::blk(w, 0));
      }
#endif
      // ---- deferred stores of the layer in front: its tiles sit in this layer's input buffer.  Where this layer's MFMA loops can host them
      // (every wave has items here and had items there) they leave in pieces between the MFMAs (WsSpread); else as one burst here.
      constexpr int MTp = R > 0 ? ws_mt<Net>(R > 0 ? R - 1 : 0) : 1;
      using ITp = WsItems<MTp>;
      constexpr int NP = R > 0 ? 4 * ITp::IPW : 0;                           // pieces per wave
      constexpr int NHOST = IT::IPW;  // (hosting only the items in front of the one that issues the next layer's A loads was tried: no difference)
      constexpr int NPI = NP / NHOST;                                        // ... per hosting item
      constexpr bool SPREAD = ST && ACTS && R > 0 && !LAST && IT::ITEMS >= 8 && ITp::ITEMS >= 8 && NP % NHOST == 0 && WsSpread<G, (NPI > 0 ? NPI : 1)>::OK;
      GLOBAL_AS void* actp = act_c;
      GLOBAL_AS unsigned int* maskp = mask_c;
      // the next layer's pointers are requested NOW (scalar loads; they have the whole layer to arrive) and become current behind the closing barrier
      const GLOBAL_AS void* Wn_n = KARG_PTR(FwdK, const void*, W, (ln + 1 < NL ? ln + 1 : 0));
      GLOBAL_AS void* act_n = KARG_PTR(FwdK, void*, act, l);
      GLOBAL_AS unsigned int* mask_n = KARG_PTR(FwdK, unsigned int*, mask, l);
      const int mtp = ITp::mt(w);
      // a wave may count itself out of block b right behind item (l, b) when nothing it does later in this layer reads block b of the input buffer:
      // two items per wave (one per block, in block order) whose hosted pieces -- the wave's own tiles of the layer in front -- belong to the item's block
      constexpr bool EARLY = MT == 8 && (R == 0 || !SPREAD || MTp == 8);
      if constexpr (R > 0 && !SPREAD) {
        auto flush_prev = [&]() {
#pragma unroll
          for (int k = 0; k < ITp::IPW; ++k) {
            const int b = ITp::blk(w, k);
            if constexpr (ACTS) {
              {
                const unsigned tb_ = xbuf_lds + (unsigned)(ib * WS_BUF * 16 + b * 2 * 16 * 1024 + mtp * 2048);
                ws_tr_issue(tb_ + trl, tb_ + trl2, trt);
              }
              tr_wait(trt);
              tr_store(actp, 32 * MTp, s0 + 64 * b, mtp, lane, trt);
            }
            if constexpr (ST && !TAN && Net::L[R > 0 ? R - 1 : 0].relu != 0) maskp[((size_t)(2 * tile + b) * MTp + mtp) * 64 + lane] = pbits[k];
          }
        };
        if constexpr (ST) {
#ifndef LAB4D_WSABL_NOFLUSH
          if (ITp::active(w)) flush_prev();
#endif
        } else {
          if (actp != nullptr && ITp::active(w)) flush_prev();  // inference: only the layer another net consumes
        }
      }
      WsPiece tp;

      if (active) {
        sfor<0, IT::IPW>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          const int b = IT::blk(w, k);
          constexpr int KL = IT::IPW - 1;
          f32x16_t acc[2];
          if constexpr (WS_SYNC || WS_TOKEN) {
            WS_T(7);
            if constexpr (WS_SYNC) {
              if (l > 0) ws_wait_block(cnt_lds + 4u * (unsigned)b, cnt_tgt);
            }
            ws_token_acquire(tok_lds, lane);
            WS_T(0);
          }
          // B units stream from LDS through a ring of WS_BD k-groups (two n-tiles each): the read of group g + WS_BD is issued right behind the
          // MFMAs of group g.  Reads and waits are volatile asm (program order kept, counted waits written by hand): left to the scheduler the
          // reads sink next to their MFMAs (lgkmcnt(1) in front of every MFMA pair) and the LDS latency is exposed once per k-group.
          constexpr int BD = G < WS_BD ? G : WS_BD;
          u32x4_t bq[BD][2];
          const unsigned pe_a = ebuf_lds + (unsigned)((2 * b * UE) * 1024 + lane * 16), px_a = xbuf_lds + (unsigned)(ib * WS_BUF * 16 + (2 * b * 16) * 1024 + lane_s * 16);
          auto b_read = [&](auto gc, u32x4_t (&dst)[2]) {
            constexpr int g = decltype(gc)::value;
#ifdef LAB4D_WSABL_HALFB  // timing experiment (results wrong): every second k-group reuses whatever the ring slot holds -- half the B reads, as if one read fed two MFMAs
            if constexpr (g % 2 == 1) return;
#endif
            if constexpr (g < GE) {
              ws_lds_read<g * 1024>(dst[0], pe_a);
              ws_lds_read<(UE + g) * 1024>(dst[1], pe_a);
            } else {
              ws_lds_read<(g - GE) * 1024>(dst[0], px_a);
              ws_lds_read<(16 + g - GE) * 1024>(dst[1], px_a);
            }
          };
          sfor<0, BD>([&](auto gc) { b_read(gc, bq[decltype(gc)::value]); });
#ifndef LAB4D_WSABL_NOBIAS
          if constexpr (k == 0) {
            if (bias_lds_path) {
              ws_lds_wait4<2 * BD>(br);  // the ring prologue (2 BD reads) was issued behind the four bias reads
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                bv[0][4 * i + 0] = __uint_as_float(br[i].x); bv[0][4 * i + 1] = __uint_as_float(br[i].y);
                bv[0][4 * i + 2] = __uint_as_float(br[i].z); bv[0][4 * i + 3] = __uint_as_float(br[i].w);
              }
              bv[1] = bv[0];
            }
          }
#endif
          acc[0] = bv[0];
          acc[1] = bv[1];
          sfor<0, G>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int NEWER = 2 * ((G - 1 - g) < (BD - 1) ? (G - 1 - g) : (BD - 1));  // reads issued behind group g's
            ws_lds_wait<NEWER>(bq[g % BD][0], bq[g % BD][1]);
            mma_b(acc[0], A[g], bq[g % BD][0]);
            mma_b(acc[1], A[g], bq[g % BD][1]);
            if constexpr (SPREAD && k < NHOST) {  // hosted pieces: global piece index k NPI + i = tile (i NP-relative) / 4 of the layer in front, quarter % 4
              using SP = WsSpread<G, (NPI > 0 ? NPI : 1)>;
              sfor<0, (SPREAD ? NPI : 0)>([&](auto ic) {
                constexpr int i = decltype(ic)::value, pi = k * NPI + i, kp = pi / 4, qa = pi % 4;
                if constexpr (SP::issue_at(i) == g) {
                  const unsigned tb_ = xbuf_lds + (unsigned)(ib * WS_BUF * 16 + ITp::blk(w, kp) * 2 * 16 * 1024 + mtp * 2048);
                  ws_trp_issue<qa>(tb_ + trl, tb_ + trl2, tp);
                }
              });
            }
            if constexpr (g + BD < G) b_read(std::integral_constant<int, g + BD>{}, bq[g % BD]);
            if constexpr (SPREAD && k < NHOST) {
              using SP = WsSpread<G, (NPI > 0 ? NPI : 1)>;
              sfor<0, (SPREAD ? NPI : 0)>([&](auto ic) {
                constexpr int i = decltype(ic)::value, pi = k * NPI + i, kp = pi / 4, qa = pi % 4;
                if constexpr (SP::perm_at(i) == g) ws_trp_perm<SP::newer(SP::issue_at(i), true, g, BD)>(perm_a, tp);
                if constexpr (SP::store_at(i) == g) {
#ifndef LAB4D_WSABL_NOFLUSH
                  ws_trp_store<qa, SP::newer(SP::perm_at(i), false, g, BD)>(actp, 32 * MTp, s0 + 64 * ITp::blk(w, kp), mtp, lane, tp);
                  if constexpr (!TAN && qa == 3 && Net::L[R > 0 ? R - 1 : 0].relu != 0) maskp[((size_t)(2 * tile + ITp::blk(w, kp)) * MTp + mtp) * 64 + lane] = pbits[kp];
#endif
                }
              });
            }
#ifndef LAB4D_WSABL_NOAPF
            if (k == KL && g < Gn) A[g] = load_a(Wn, Gn, mtn, g, lane);  // the next layer's group g, right behind the last use of this one
#endif
          });
          ws_token_release(tok_lds, lane);
          WS_T(1 + 2 * (k & 1));
          // requests of the next item (they have that item's matrix work to arrive)
          if (k < KL) {
#ifndef LAB4D_WSABL_NOBIAS
            if constexpr (ls.pf != 0) load_bias(IT::blk(w, k + 1), bv);
#endif
          }
          // ---- epilogue ----
          if constexpr (!LAST && ls.add_ext == 0) {
            // packed-bf16 epilogue (pk_* helpers of mlp_kernels.hpp)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc_fence(acc[t]);
            unsigned int pw[2][8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int kk = 0; kk < 8; ++kk) pw[t][kk] = pack2bf_op(acc[t][2 * kk], acc[t][2 * kk + 1]);
            if constexpr (ls.relu != 0 && TAN) {
#pragma unroll
              for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) pw[t][kk] = pk_mask_bf16(pw[t][kk], pk_m01(tbits[k], t, kk));
            } else if constexpr (ls.relu != 0) {
              if constexpr (ST) pbits[k] = pk_alive_bits(pw);
#pragma unroll
              for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) pw[t][kk] = pk_relu_bf16(pw[t][kk]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int q = 0; q < 2; ++q)
                xout[((2 * b + t) * 16 + 2 * mt + q) * 64 + lane_s] = make_uint4(pw[t][4 * q], pw[t][4 * q + 1], pw[t][4 * q + 2], pw[t][4 * q + 3]);
          } else if constexpr (!LAST) {
            // layer with an external add (colour net: + basefield feature): fp32 ReLU, sign word by comparison, + ext, then the units
            if constexpr (ls.relu != 0) {
              if constexpr (ST) {
                unsigned int bits = 0;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                  for (int r = 0; r < 16; ++r) bits |= (acc[t][r] > 0.f ? 1u : 0u) << mask_bit<P>(t, r);
                pbits[k] = bits;
              }
#pragma unroll
              for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = relu1(acc[t][r]);
            }
            {
              f32x16_t e[2];
              tile_from_raw<P>(ext_raw, lane, e);
#pragma unroll
              for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += e[t][r];
            }
            if (k < KL) load_ext(IT::blk(w, k + 1));
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              uint4 u[2];
              tile_to_units<P>(acc[t], u);
#pragma unroll
              for (int q = 0; q < 2; ++q) xout[((2 * b + t) * 16 + 2 * mt + q) * 64 + lane_s] = u[q];
            }
          } else {
            // head: raw outputs (S, COUT) fp32
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int s = s0 + 64 * b + 2 * n + t;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int f = 32 * mt + drow(r, h);
                if (f < Net::COUT && s < S_eff && a.out) a.out[(size_t)s * Net::COUT + f] = acc[t][r];
              }
            }
          }
          if constexpr (WS_SYNC && EARLY && !LAST) ws_arrive(cnt_lds + 4u * (unsigned)b, lane);
          WS_T(2 + 2 * (k & 1));
        });
      } else {
        // a wave without an item in this layer still needs the next layer's weights
        a_load(std::integral_constant<int, 0>{}, std::integral_constant<int, (Gn < G ? Gn : G)>{}, Wn, Gn, mtn);
        if constexpr (LAST && HIDE) {
          static_assert(!LAST || !HIDE || IT::ITEMS == 2, "HIDE: the head's two items belong to waves 0 and 1");
#ifndef LAB4D_WSABL_NOPOSENC
          if (tile + (int)gridDim.x < ntw) posenc_rows(xnext);  // the NEXT tile's positional encoding, under the head's matrix work (see HIDE)
#endif
        }
      }
      if constexpr (WS_SYNC && !LAST) {
        if (!(EARLY && active)) {  // counted out of both blocks at the end of the layer
          // ... and not before the layer in front is complete on BOTH: a wave with one item (or none) never waited on the other block, and its
          // arrival there would be taken for a missing one of the layer in front (the counters are sums; found on hardware: fg_color's 4-row-tile layer)
          if (l > 0) {
            ws_wait_block(cnt_lds, cnt_tgt);
            ws_wait_block(cnt_lds + 4u, cnt_tgt);
          }
          ws_arrive(cnt_lds, lane);
          ws_arrive(cnt_lds + 4u, lane);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // the scalar loads of the next layer's pointers (see ws_layer_barrier)
      } else {
        ws_layer_barrier();  // tile boundary (and every layer with LAB4D_WS_SYNC=0)
      }
      Wn_c = Wn_n;
      act_c = act_n;
      mask_c = mask_n;
      ws_pin_sgpr(Wn_c);
      ws_pin_sgpr(act_c);
      ws_pin_sgpr(mask_c);
      WS_T(5);
    });
    cnt_base += 8u * (unsigned)(NL - 1);  // NL - 1 counted layer ends per tile (the last layer ends at the tile's barrier)
    xcur = xnext;
  }
#ifdef LAB4D_WS_TRACE
  if (blockIdx.x == 0 && lane < 8 && a.out) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) v = lane == i ? tacc[i] : v;
    a.out[w * 8 + lane] = v;
  }
#endif
}

// =================================================================================================
// backward (dgrad) chain, weights stationary
// =================================================================================================
// Same dataflow with W^T: wave w keeps row tile (MTE + w) of W^T[l] (32 features of layer l-1's output x the K = mout_pad contraction) in
// registers, dZ_l of all 128 samples streams from LDS, the masked dZ_{l-1} slice goes to the other buffer.  The embedding row tiles of W^T
// (layers that consume the posenc: input gradient) are 2 MTE more items, taken by waves 0 .. 2 MTE - 1 in front of their activation items.
template <class Net>
constexpr int wsb_gk(int l) { return pad32(Net::L[l].mout) / 16; }
template <class Net>
constexpr bool wsb_same(int a, int b) {
  if (!bwd_same<Net>(a, b)) return false;
  if ((a == Net::NL - 1) != (b == Net::NL - 1)) return false;
  const int an = a > 0 ? a - 1 : Net::NL - 1, bn = b > 0 ? b - 1 : Net::NL - 1;
  return wsb_gk<Net>(an) == wsb_gk<Net>(bn);
}
template <class Net>
constexpr int wsb_rep(int l) {
  for (int j = Net::NL - 1; j > l; --j)
    if (wsb_same<Net>(j, l)) return j;
  return l;
}
template <class Net>
constexpr unsigned wsb_members(int r) {
  unsigned m = 0;
  for (int l = 0; l < Net::NL; ++l)
    if (wsb_rep<Net>(l) == r) m |= 1u << l;
  return m;
}
// row tile of W^T[l] wave w starts layer l with: its embedding item's if it has one, else its activation item's
template <class Net>
__device__ __forceinline__ int wsb_first_tile(int l, int w, bool want_dx) {
  int v = 0;
  sfor<0, Net::NL>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int MTE = Net::L[i].ke / 32, MTA = Net::L[i].kin / 32;
    if (l == i) {
      if (MTE > 0 && want_dx && w < 2 * MTE) v = w % (MTE > 0 ? MTE : 1);
      else v = (i > 0 && MTA > 0) ? MTE + (w & ((MTA > 0 ? MTA : 1) - 1)) : 0;  // (a layer without activation rows: any valid tile, the fetch is ignored)
    }
  });
  return v;
}
// row tiles of the activation part of layer l (0: none), runtime
template <class Net>
__device__ __forceinline__ int wsb_mta_rt(int l) {
  int v = 0;
  sfor<0, Net::NL>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if (l == i) v = i > 0 ? Net::L[i].kin / 32 : 0;
  });
  return v;
}

template <class Net, bool DZ = true>
__global__ void __launch_bounds__(512) k_mlp_bwd_ws(BwdK a) {
  static_assert(ws_ok<Net>(), "weights-stationary chain: 256-wide posenc nets only");
  using P = PBF16;
  constexpr int NL = Net::NL, KE = Net::KE, GMAX = 16;
  constexpr int MTE_ANY = KE / 32;
  __shared__ uint4 xbuf[2 * WS_BUF];
  __shared__ float red[8 * 2 * 3 * 32];  // input-gradient partials of the embedding items: [wave][n-tile][axis][lane n]
  __shared__ unsigned int blk_cnt[2 + 4];  // per-block progress counters (see ws_arrive) + per-SIMD matrix-pipe tokens
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned xbuf_lds = lds_addr(xbuf), cnt_lds = lds_addr(blk_cnt), tok_lds = cnt_lds + 4u * (2u + (unsigned)(w & 3));
  if constexpr (WS_SYNC || WS_TOKEN) {
    if (tid < 6) blk_cnt[tid] = 0u;
  }
  unsigned cnt_base = 0u;
  const unsigned trl = ws_tr_lane(lane), trl2 = trl ^ 32u;  // (see the forward kernel: slot swizzle of the slabs)
  const int lane_s = ws_slot(lane);
  const unsigned perm_a = ws_perm_addr(lane);
  const int ntw = a.S_pad / WS_TILE;
  const bool want_dx = a.d_x != nullptr;

  uint4 A[GMAX];
  auto a_load = [&](auto g1c, const GLOBAL_AS void* Wp, int G, int rt) {
    constexpr int G1 = decltype(g1c)::value;
#pragma unroll
    for (int g = 0; g < G1; ++g) A[g] = load_a(Wp, G, rt, g, lane);
  };
  if ((int)blockIdx.x < ntw) {
    constexpr int GK0 = wsb_gk<Net>(NL - 1);
    a_load(std::integral_constant<int, GK0>{}, KARG_PTR(BwdK, const void*, WT, NL - 1), GK0, wsb_first_tile<Net>(NL - 1, w, want_dx));
  }
  TrTile trt;
  // pointers the NEXT layer needs (see the forward kernel): the weights of the layer below it, its own dZ buffer, the sign words its successor's items want
  auto mask_slot_after = [](int ln_) { const int q = ln_ > 0 ? ln_ - 1 : 0; return q >= 1 ? q - 1 : 0; };
  const GLOBAL_AS void* Wn_c = KARG_PTR(BwdK, const void*, WT, (NL >= 2 ? NL - 2 : 0));
  GLOBAL_AS void* dz_c = KARG_PTR(BwdK, void*, dz, NL - 1);
  const GLOBAL_AS unsigned int* maskq_c = KARG_PTR(BwdK, const unsigned int*, mask, mask_slot_after(NL - 1));
  // ReLU sign words of this wave's activation items: requested one layer ahead (they come from HBM, written a whole chunk earlier)
  auto mask_req = [&](int lq /* layer whose activation items want them: words of mask[lq - 1] */, int tile, unsigned int (&m)[2], const GLOBAL_AS unsigned int* mp) {
    int mta = wsb_mta_rt<Net>(lq >= 1 ? lq : 1);
    mta = mta > 0 ? mta : 8;
    const int j = w & (mta - 1);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int b = mta >= 8 ? k : ((w / mta) & 1);
      m[k] = mp[((size_t)(2 * tile + b) * mta + j) * 64 + lane];
    }
  };

  // Round 5 (see the forward kernel): the head gradient of the NEXT tile is requested a tile ahead (heads of up to four outputs: one to four floats per
  // lane of waves 0 / 1) -- the tile no longer opens with an HBM round trip in front of its first barrier.
  constexpr bool DPRE = Net::COUT <= 4;
  constexpr int DN = DPRE ? Net::COUT : 1;
  auto d_fetch = [&](int tile_, float (&d)[2][DN]) {
    if constexpr (DPRE) {
      if (w < 2 && tile_ < ntw) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int s = tile_ * WS_TILE + 64 * w + 2 * n + t;
#pragma unroll
          for (int r = 0; r < DN; ++r) {
            const int f = drow(r, h);  // (= r for the lane half h = 0, >= 4 for h = 1)
            d[t][r] = (f < Net::COUT && s < a.S) ? a.d_out[(size_t)s * Net::COUT + f] : 0.f;
          }
        }
      }
    }
  };
  float dcur[2][DN], dnext[2][DN];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < DN; ++r) dcur[t][r] = dnext[t][r] = 0.f;
  d_fetch((int)blockIdx.x, dcur);
  for (int tile = blockIdx.x; tile < ntw; tile += gridDim.x) {
    const int s0 = tile * WS_TILE;
    d_fetch(tile + (int)gridDim.x, dnext);
    float dx[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t) dx[t][0] = dx[t][1] = dx[t][2] = 0.f;
    unsigned int mcur[2], mnext[2];
    mask_req(NL - 1, tile, mcur, KARG_PTR(BwdK, const unsigned int*, mask, (NL >= 2 ? NL - 2 : 0)));

    // ---- head gradient: (S, COUT) fp32 -> accumulator layout -> stored + B units of buffer 0 (waves 0, 1: one 64-sample block each) ----
    if (w < 2) {
      const int b = w;
      f32x16_t g[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int s = s0 + 64 * b + 2 * n + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = drow(r, h);
          if constexpr (DPRE) g[t][r] = r < DN ? dcur[t][r < DN ? r : 0] : 0.f;  // (registers r >= 4 hold features >= 8 > COUT)
          else g[t][r] = (f < Net::COUT && s < a.S) ? a.d_out[(size_t)s * Net::COUT + f] : 0.f;
        }
      }
      if constexpr (DZ) store_tile<P>((GLOBAL_AS void*)a.dz[NL - 1], pad32(Net::L[NL - 1].mout), s0 + 64 * b, 0, lane, g);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        uint4 u[2];
        tile_to_units<P>(g[t], u);
#pragma unroll
        for (int q = 0; q < 2; ++q) xbuf[((2 * b + t) * 16 + q) * 64 + lane_s] = u[q];
      }
    }
    wg_step_barrier();

#pragma nounroll
    for (int l = NL - 1; l >= 0; --l)
    sfor<0, NL>([&](auto ri) {
      constexpr int R = NL - 1 - decltype(ri)::value;
      if constexpr (wsb_rep<Net>(R) != R) return;
      constexpr unsigned MEMBERS = wsb_members<Net>(R);
      if (!((MEMBERS >> l) & 1u)) return;
      constexpr LS ls = Net::L[R];
      constexpr LS lp = Net::L[R > 0 ? R - 1 : 0];
      constexpr int GK = wsb_gk<Net>(R);
      constexpr int MTE = ls.ke / 32, MTA = ls.kin / 32;
      constexpr bool DO_ACT = (R > 0 && MTA > 0);
      constexpr int GKn = wsb_gk<Net>(R > 0 ? R - 1 : NL - 1);  // K groups of the layer below (the top layer of the next tile below layer 0)
      using IT = WsItems<(DO_ACT ? MTA : 1)>;
      const int lm1 = l > 0 ? l - 1 : 0;
      const int ln = l > 0 ? l - 1 : NL - 1;
      const GLOBAL_AS void* Wn = Wn_c;
      // the next layer's pointers are requested now and become current behind the closing barrier
      const GLOBAL_AS void* Wn_n = KARG_PTR(BwdK, const void*, WT, (ln > 0 ? ln - 1 : NL - 1));
      GLOBAL_AS void* dz_n = KARG_PTR(BwdK, void*, dz, ln);
      const GLOBAL_AS unsigned int* maskq_n = KARG_PTR(BwdK, const unsigned int*, mask, mask_slot_after(ln));
      const int rtn = wsb_first_tile<Net>(ln, w, want_dx);
      const int ib = (NL - 1 - l) & 1;
      uint4* xout = xbuf + (ib ^ 1) * WS_BUF;
      const unsigned cnt_tgt = cnt_base + 8u * (unsigned)(NL - 1 - l);  // every wave has finished the layer above on a block once its counter reads this
      const bool act_on = DO_ACT && IT::active(w);
      const bool emb_on = MTE > 0 && want_dx && w < 2 * MTE;
      const int j = IT::mt(w);              // activation row tile of this wave
      const int me = MTE > 0 ? w % (MTE > 0 ? MTE : 1) : 0, be = MTE > 0 ? (w / (MTE > 0 ? MTE : 1)) & 1 : 0;  // embedding item

      // ---- requests first: the next layer's A groups this layer has no use for, its sign words, this layer's stored embedding / external gradient tile ----
#pragma unroll
      for (int g = GK; g < GKn; ++g) A[g] = load_a(Wn, GKn, rtn, g, lane);
      mask_req(lm1, tile, mnext, maskq_c);
      uint4 raw[4];
      if constexpr (MTE > 0) {
        if (emb_on) load_tile_raw<P>((const GLOBAL_AS void*)a.emb, KE, s0 + 64 * be, me, lane, raw);
      }
      if constexpr (DO_ACT && lp.ext_grad != 0) {
        if (act_on && !emb_on) load_tile_raw<P>((const GLOBAL_AS void*)a.ext_gin, pad32(lp.mout), s0 + 64 * IT::blk(w, 0), j, lane, raw);
      }
      static_assert(!(MTE > 0 && DO_ACT && lp.ext_grad != 0), "one prefetched tile per layer");
      // ---- deferred stores: dZ_l (this layer's input, written by the layer above): in pieces between the MFMAs of the activation items where
      // they can host them (WsSpread), else as one burst here ----
      constexpr int MTp = GK / 2;
      using ITp = WsItems<MTp>;
      constexpr int NP = (DZ && R < NL - 1) ? 4 * ITp::IPW : 0;
      constexpr int NHOST = IT::IPW;
      constexpr int NPI = NP / NHOST;
      constexpr bool SPREAD = NP > 0 && DO_ACT && IT::ITEMS >= 8 && ITp::ITEMS >= 8 && NP % NHOST == 0 && WsSpread<GK, (NPI > 0 ? NPI : 1)>::OK;
      GLOBAL_AS void* dzl = dz_c;
      const int jp = ITp::mt(w);
      constexpr bool EARLY = DO_ACT && MTA == 8 && (NP == 0 || !SPREAD || MTp == 8);  // (see the forward kernel)
      constexpr bool TOP = R == NL - 1, BOTTOM = R == 0;
      if constexpr (NP > 0 && !SPREAD) {
        if (ITp::active(w)) {
#pragma unroll
          for (int k = 0; k < ITp::IPW; ++k) {
            const int b = ITp::blk(w, k);
            {
              const unsigned tb_ = xbuf_lds + (unsigned)(ib * WS_BUF * 16 + b * 2 * 16 * 1024 + jp * 2048);
              ws_tr_issue(tb_ + trl, tb_ + trl2, trt);
            }
            tr_wait(trt);
            tr_store(dzl, 32 * MTp, s0 + 64 * b, jp, lane, trt);
          }
        }
      }
      WsPiece tp;

      // one item: acc = W^T[row tile in A] dZ_l over block b; PF: the groups of (Wp, GP groups, row tile rt) replace A behind their last use
      auto mfma_item = [&](int b, f32x16_t (&acc)[2], auto pf_c, auto gp_c, const GLOBAL_AS void* Wp, int rt, auto host_c) {
        constexpr bool PF = decltype(pf_c)::value;
        constexpr int GP = decltype(gp_c)::value;
        constexpr int HK = decltype(host_c)::value;  // >= 0: this loop hosts the pieces HK NPI .. of the deferred dZ tiles
        if constexpr (WS_SYNC && !TOP) ws_wait_block(cnt_lds + 4u * (unsigned)b, cnt_tgt);  // (the top layer's input is staged in front of the tile's barrier)
        ws_token_acquire(tok_lds, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        constexpr int BD = GK < WS_BD ? GK : WS_BD;
        u32x4_t bq[BD][2];
        const unsigned px_a = xbuf_lds + (unsigned)(ib * WS_BUF * 16 + (2 * b * 16) * 1024 + lane_s * 16);
        auto b_read = [&](auto gc, u32x4_t (&dst)[2]) {
          constexpr int g = decltype(gc)::value;
#ifdef LAB4D_WSABL_HALFB
          if constexpr (g % 2 == 1) return;
#endif
          ws_lds_read<g * 1024>(dst[0], px_a);
          ws_lds_read<(16 + g) * 1024>(dst[1], px_a);
        };
        sfor<0, BD>([&](auto gc) { b_read(gc, bq[decltype(gc)::value]); });
        sfor<0, GK>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          constexpr int NEWER = 2 * ((GK - 1 - g) < (BD - 1) ? (GK - 1 - g) : (BD - 1));
          ws_lds_wait<NEWER>(bq[g % BD][0], bq[g % BD][1]);
          mma_b(acc[0], A[g], bq[g % BD][0]);
          mma_b(acc[1], A[g], bq[g % BD][1]);
          if constexpr (SPREAD && HK >= 0 && HK < NHOST) {
            using SP = WsSpread<GK, (NPI > 0 ? NPI : 1)>;
            sfor<0, (SPREAD ? NPI : 0)>([&](auto ic) {
              constexpr int i = decltype(ic)::value, pi = (HK >= 0 ? HK : 0) * NPI + i, kp = pi / 4, qa = pi % 4;
              if constexpr (SP::issue_at(i) == g) {
                const unsigned tb_ = xbuf_lds + (unsigned)(ib * WS_BUF * 16 + ITp::blk(w, kp) * 2 * 16 * 1024 + jp * 2048);
                ws_trp_issue<qa>(tb_ + trl, tb_ + trl2, tp);
              }
            });
          }
          if constexpr (g + BD < GK) b_read(std::integral_constant<int, g + BD>{}, bq[g % BD]);
          if constexpr (SPREAD && HK >= 0 && HK < NHOST) {
            using SP = WsSpread<GK, (NPI > 0 ? NPI : 1)>;
            sfor<0, (SPREAD ? NPI : 0)>([&](auto ic) {
              constexpr int i = decltype(ic)::value, pi = (HK >= 0 ? HK : 0) * NPI + i, kp = pi / 4, qa = pi % 4;
              if constexpr (SP::perm_at(i) == g) ws_trp_perm<SP::newer(SP::issue_at(i), true, g, BD)>(perm_a, tp);
              if constexpr (SP::store_at(i) == g) ws_trp_store<qa, SP::newer(SP::perm_at(i), false, g, BD)>(dzl, 32 * MTp, s0 + 64 * ITp::blk(w, kp), jp, lane, tp);
            });
          }
          if constexpr (PF && g < GP) A[g] = load_a(Wp, GP, rt, g, lane);
        });
        ws_token_release(tok_lds, lane);
      };

      bool loaded_next = false;
      // ---- (a) embedding rows: gradient wrt the posenc slots -> input gradient ----
      if constexpr (MTE > 0) {
        if (emb_on) {
          f32x16_t acc[2];
          if constexpr (DO_ACT) mfma_item(be, acc, std::true_type{}, std::integral_constant<int, GK>{}, KARG_PTR(BwdK, const void*, WT, l), MTE + j, std::integral_constant<int, -1>{});
          else { mfma_item(be, acc, std::true_type{}, std::integral_constant<int, GKn>{}, Wn, rtn, std::integral_constant<int, -1>{}); loaded_next = true; }
          f32x16_t e[2];
          tile_from_raw<P>(raw, lane, e);
          constexpr int L = Net::NFREQ;
          // Round 5: the embedding row tile `me` is resolved at COMPILE time (one copy of this block per tile, selected by a wave-uniform branch).  With a
          // runtime `me` every accumulator register's slot -> (octave, axis, class) map was loop-invariant scalar state: the compiler hoisted ~135 values into
          // SGPRs, spilled them to VGPR lanes (177-203 spilled SGPRs in these kernels) and read them back with v_readlane in front of every
          // use -- on the waves that carry the layer's critical path (they run the embedding item in front of their activation items).  Now the maps are
          // immediates; per register and lane half only the octave (an exponent for ldexpf) and the axis select remain.  Same products, same order of
          // the additions into dx per axis; additions of +0 are no longer performed.
          sfor<0, (MTE > 0 ? MTE : 1)>([&](auto mc) {
            constexpr int MEc = decltype(mc)::value;
            if (me != MEc) return;
            sfor<0, 16>([&](auto rc) {
              constexpr int r = decltype(rc)::value;
              constexpr int S0 = 32 * MEc + (r & 3) + 8 * (r >> 2), S1 = S0 + 4;  // slot of this register in lane half 0 / 1 (drow)
              constexpr int K0 = S0 < 6 * L ? 1 : (S0 < 6 * L + 3 ? 2 : 0), K1 = S1 < 6 * L ? 1 : (S1 < 6 * L + 3 ? 2 : 0);  // 1: sin / cos slot, 2: raw coordinate, 0: padding
              if constexpr (K0 != 0 || K1 != 0) {
                constexpr int F0 = (S0 >> 1) / 3, F1 = (S1 >> 1) / 3;
                constexpr int A0 = K0 == 1 ? (S0 >> 1) % 3 : (K0 == 2 ? S0 - 6 * L : -1), A1 = K1 == 1 ? (S1 >> 1) % 3 : (K1 == 2 ? S1 - 6 * L : -1);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                  const float gv = acc[t][r];
                  const float partner = e[t][r ^ 1];
                  const float sp = (r & 1) ? -partner : partner;  // d/dx [w sin(2^f x)] = 2^f (w cos) ; d/dx [w cos(2^f x)] = -2^f (w sin)
                  float c;
                  if constexpr (K0 == 1 && K1 == 1) {
                    c = ldexpf(sp, h ? F1 : F0) * gv;
                  } else {
                    const float c0 = K0 == 1 ? ldexpf(sp, F0) * gv : (K0 == 2 ? gv : 0.f);
                    const float c1 = K1 == 1 ? ldexpf(sp, F1) * gv : (K1 == 2 ? gv : 0.f);
                    c = h ? c1 : c0;
                  }
#pragma unroll
                  for (int k = 0; k < 3; ++k) {
                    if (A0 == k && A1 == k) dx[t][k] += c;
                    else if (A0 == k) dx[t][k] += h ? 0.f : c;
                    else if (A1 == k) dx[t][k] += h ? c : 0.f;
                  }
                }
              }
            });
          });
        }
      }
      // ---- (b) activation rows: masked dZ_{l-1} ----
      if constexpr (DO_ACT) {
        if (act_on) {
          sfor<0, IT::IPW>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const int b = IT::blk(w, k);
            constexpr int KL = IT::IPW - 1;
              f32x16_t acc[2];
            if constexpr (k == KL) mfma_item(b, acc, std::true_type{}, std::integral_constant<int, GKn>{}, Wn, rtn, kc);
            else mfma_item(b, acc, std::false_type{}, std::integral_constant<int, 0>{}, Wn, rtn, kc);
            if constexpr (lp.ext_grad != 0) {
              f32x16_t eg[2];
              tile_from_raw<P>(raw, lane, eg);
#pragma unroll
              for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += eg[t][r];
              if (k < KL) load_tile_raw<P>((const GLOBAL_AS void*)a.ext_gin, pad32(lp.mout), s0 + 64 * IT::blk(w, k + 1), j, lane, raw);
            }
            if constexpr (lp.add_ext != 0) store_tile<P>((GLOBAL_AS void*)a.ext_gout, pad32(lp.mout), s0 + 64 * b, j, lane, acc);  // y = relu(z) + ext -> dL/dext = dL/dy
#pragma unroll
            for (int t = 0; t < 2; ++t) acc_fence(acc[t]);
            unsigned int pw[2][8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int kk = 0; kk < 8; ++kk) pw[t][kk] = pack2bf_op(acc[t][2 * kk], acc[t][2 * kk + 1]);
            {
              unsigned int alive = lp.relu != 0 ? mcur[k] : 0xffffffffu;
#pragma unroll
              for (int t = 0; t < 2; ++t)
                if (s0 + 64 * b + 2 * n + t >= a.S) alive &= ~(0x00ff00ffu << (8 * t));
#pragma unroll
              for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) pw[t][kk] = pk_mask_bf16(pw[t][kk], pk_m01(alive, t, kk));
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int q = 0; q < 2; ++q)
                xout[((2 * b + t) * 16 + 2 * j + q) * 64 + lane_s] = make_uint4(pw[t][4 * q], pw[t][4 * q + 1], pw[t][4 * q + 2], pw[t][4 * q + 3]);
            if constexpr (WS_SYNC && EARLY && !BOTTOM) ws_arrive(cnt_lds + 4u * (unsigned)b, lane);
          });
          loaded_next = true;
        }
      }
      if (!loaded_next) a_load(std::integral_constant<int, (GKn < GK ? GKn : GK)>{}, Wn, GKn, rtn);  // a wave without an item here still needs its next weights
      mcur[0] = mnext[0];
      mcur[1] = mnext[1];
      if constexpr (WS_SYNC && !BOTTOM) {
        if (!(EARLY && act_on)) {
          if constexpr (!TOP) {  // (see the forward kernel: no arrival on a block before the layer above is complete there)
            ws_wait_block(cnt_lds, cnt_tgt);
            ws_wait_block(cnt_lds + 4u, cnt_tgt);
          }
          ws_arrive(cnt_lds, lane);
          ws_arrive(cnt_lds + 4u, lane);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
      } else {
        ws_layer_barrier();  // tile boundary (and every layer with LAB4D_WS_SYNC=0)
      }
      Wn_c = Wn_n;
      dz_c = dz_n;
      maskq_c = maskq_n;
      ws_pin_sgpr(Wn_c);
      ws_pin_sgpr(dz_c);
      ws_pin_sgpr(maskq_c);
    });

    cnt_base += 8u * (unsigned)(NL - 1);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < DN; ++r) dcur[t][r] = dnext[t][r];
    // ---- input gradient: partials of the embedding items -> one sum per sample ----
    if (want_dx) {
      if (w < 2 * MTE_ANY) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float v = dx[t][k] + __shfl_xor(dx[t][k], 32, 64);
            if (h == 0) red[((w * 2 + t) * 3 + k) * 32 + n] = v;
          }
      }
      wg_step_barrier();
      if (w < 2 && h == 0) {  // wave b sums the MTE partials of block b
        const int b = w;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int s = s0 + 64 * b + 2 * n + t;
          float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
          for (int m = 0; m < MTE_ANY; ++m)
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] += red[(((b * MTE_ANY + m) * 2 + t) * 3 + k) * 32 + n];
          if (s < a.S) {
            a.d_x[(size_t)s * 3 + 0] = v[0];
            a.d_x[(size_t)s * 3 + 1] = v[1];
            a.d_x[(size_t)s * 3 + 2] = v[2];
          }
        }
      }
    }
  }
}

// 8-wave persistent grid: one workgroup per CU
inline int mlp_grid_ws(int ntiles) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  // read per launch (like LAB4D_WS): tests/test_gpu_mlp_ws.py forces a 3-workgroup grid so that every workgroup runs several tiles of a small case
  const char* ge = getenv("LAB4D_CHAIN_GRID");
  const int grid_env = ge ? atoi(ge) : 0;
  int g = ntiles < n_cu ? ntiles : n_cu;
  if (grid_env > 0 && g > grid_env) g = grid_env;
  return g < 1 ? 1 : g;
}

// LAB4D_WS=0 routes the 256-wide nets back to the wave-resident kernels (read per launch: the tests run both families in one process)
inline bool ws_enabled() {
  const char* e = getenv("LAB4D_WS");
  return e == nullptr || atoi(e) != 0;
}

}  // namespace lab4d
