// Fused positional-encoding + MLP stack for gfx950: forward chain and backward (dgrad) chain
// kernel templates.  Instantiated per network in mlp_inst_*.hip (one translation unit each, so the
// nets compile in parallel); host API in mlp.hip.  Contract: include/lab4d_mlp.h.
//
// Wave-resident chain.  A wavefront owns TILE = 32*NT samples (NT = 2 for bf16, 1 for fp32) and
// carries them through ALL layers without ever exchanging data with another wave:
// lane l = (n = l & 31, h = l >> 5) owns sample column n of each 32-sample n-tile.  For the 32x32
// MFMA shapes used here (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32):
//   * accumulator D:  lane (n,h), register r  <->  output feature  32*mt + (r&3) + 8*(r>>2) + 4*h
//   * B operand:      lane (n,h) supplies k-slots {8h..8h+7} (bf16) or {h} (fp32) of column n
// The contraction over k does not care about the order of k as long as A and B agree, so the packed
// A-fragments of layer l+1 are laid out such that "k-slot (step, h, j)" means exactly the feature a
// lane already holds in accumulator register (step, j) of layer l: the accumulator IS the next B
// operand (after bias/ReLU/convert) -- no transposition, no cross-lane traffic, no barrier.
// (Verified on hardware by tests/test_gpu_ops.py::test_mfma_layout_probe.)
//
// A layer's inputs live in registers as 16-byte "units" (one A-group's worth of B data: 8 bf16 or
// 4 fp32 k-steps).  Outputs are produced M-tile by M-tile in a *runtime* loop and parked in a
// wave-private LDS slab addressed [n-tile][unit][lane] -- every lane reads and writes only its own
// 16-byte slots (lane-linear ds_read/write_b128, conflict-free, no synchronisation); the slab is
// just dynamically-indexable register space.  The next layer reloads its inputs from the slab.
// Weights stream from L2 as pre-packed, lane-linear 1-KiB A-groups (global_load_dwordx4).
#pragma once
#include <cstddef>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "mlp_nets.hpp"

namespace lab4d {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct PBF16 {
  static constexpr bool BF16 = true;
  static constexpr int NT = 2, TILE = 64, FPG = 16, UPT = 2;  // features per 16-byte unit, units per 32-feature tile
  using store_t = unsigned short;
};
struct PF32 {
  static constexpr bool BF16 = false;
  static constexpr int NT = 1, TILE = 32, FPG = 8, UPT = 4;
  using store_t = float;
};

__device__ __forceinline__ unsigned short f2bf(float x) {  // round to nearest even; NaN stays NaN
  unsigned int u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }
// two fp32 -> packed bf16x2 with ONE v_cvt_pk_bf16_f32 (round-to-nearest-even in hardware).  The bit-twiddling
// f2bf above costs ~7 VALU per value; with 64 conversions per M-tile it was as expensive as the tile's MFMAs.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) {
  const bf16x2_hw v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned int, v);
}

// One octave of the angle-doubling recurrence of the bf16 posenc (sin 2a = 2 s c, cos 2a = 1 - 2 s^2) with its roundings pinned: 2 s is exact, one
// multiply, one fma.  Written as plain arithmetic the optimiser picks per call site between the fma and a packed multiply + add (v_pk_mul_f32 /
// v_pk_add_f32 once the SLP vectoriser pairs the two lines), and two kernels evaluating the same samples disagreed in the last bit of 0.1 % of the
// stored embedding (found round 4 when the weights-stationary kernels were held bit-equal to these).
__device__ __forceinline__ void sincos_double(float& sn, float& cs) {
  const float t = sn + sn;
  const float s2 = t * cs, c2 = __builtin_fmaf(-t, sn, 1.f);
  sn = s2;
  cs = c2;
}

// max(x, 0) in ONE v_max_f32: fmaxf() first canonicalises its argument (a second v_max) to quiet signalling NaNs, which an
// MFMA result never is
__device__ __forceinline__ float relu1(float x) {
  // one v_med3_f32 the compiler can see (its hazard recognizer then places the wait states an MFMA result needs in front of a VALU
  // read; the inline-asm v_max it replaces was invisible to it, see acc_fence)
  return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff());
}

// Hazard fence between the MFMAs that produced an accumulator tile and INLINE ASM that reads it.  On gfx950 a VALU read of an
// MFMA result needs software wait states (passes + 3: 11 for the 8-pass 32x32x16, 19 for a 16-pass shape); the compiler inserts
// them in front of instructions it can see, but not in front of an inline-asm statement.  While the accumulators lived in AGPRs a
// compiler-visible v_accvgpr_read always sat in between; built for two waves per SIMD (<= 256 registers, no AGPRs) the packed
// epilogue read the MFMA's destination VGPRs directly and, depending on the schedule, got stale values (found on hardware:
// visibility / feature nets wrong at occupancy 2, right at occupancy 1).  The fence takes the tile as in/out operands, so every
// asm consumer depends on it; 20 wait states, only ever exposed when the tile's last MFMA has just issued.
__device__ __forceinline__ void acc_fence(f32x16_t& c) {
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3"
               : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]),
                 "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]));
}

// Kernels built for ONE wave per SIMD use more than 256 registers, so their accumulators live in AGPRs and a compiler-visible v_accvgpr_read
// (whose hazards the compiler handles) always sits between the MFMA and the asm consumer: the 40 NOP states per step are only needed by the
// two-waves-per-SIMD builds, or when the MFMA results are forced into VGPRs (-mllvm -amdgpu-mfma-vgpr-form: define LAB4D_MFMA_VGPR_FORM).
template <bool OCC2>
__device__ __forceinline__ void acc_fence_if(f32x16_t& c) {
#if defined(LAB4D_FENCE_ALWAYS) || defined(LAB4D_MFMA_VGPR_FORM)
  acc_fence(c);
#else
  if constexpr (OCC2) acc_fence(c);  // round 4: -2 % of the step together with the scheduler flag of _lib.MLP_INST_FLAGS (profiles/r04_flag_variants.json)
#endif
}

// feature (row) held by accumulator register r of lane-half h inside a 32-row tile
__host__ __device__ __forceinline__ constexpr int drow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

#define GLOBAL_AS __attribute__((address_space(1)))
// plain clang vectors for accesses through address-space-qualified pointers (HIP's uint4/float4 are classes whose
// copy constructors only bind generic references)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 gld16(const GLOBAL_AS void* p) {
  const u32x4_t v = *(const GLOBAL_AS u32x4_t*)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gst16(GLOBAL_AS void* p, unsigned int x, unsigned int y, unsigned int z, unsigned int w) {
  u32x4_t v = {x, y, z, w};
  // streaming (nt) store: stored activations / dZ / embeddings are written once and read once, much later, by the weight-gradient
  // kernels -- measured -6.5 % on the training-mode forward and -6.3 % on the backward chain against plain stores
#ifdef LAB4D_ABL_PLAINSTORE
  *(GLOBAL_AS u32x4_t*)p = v;
#elif defined(LAB4D_ST_AGPR)  // kernel experiment: the store takes its data from accumulation registers
  asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "a"(v) : "memory");
#else
  __builtin_nontemporal_store(v, (GLOBAL_AS u32x4_t*)p);
#endif
}
// the same store as (wave-uniform tile base, per-lane byte offset): -DLAB4D_ST_BUF issues it as buffer_store_dwordx4 with the base in an
// SGPR resource descriptor (kernel experiment); otherwise identical to gst16(base + off, ..)
__device__ __forceinline__ void gst16o(GLOBAL_AS char* base, unsigned off, unsigned int x, unsigned int y, unsigned int z, unsigned int w) {
#ifdef LAB4D_ST_BUF
  u32x4_t v = {x, y, z, w};
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)off, 0, 2);
#else
  gst16(base + off, x, y, z, w);
#endif
}

// 16-byte A group load: the packed block of (mt, g) is 1 KiB, lane-linear.
// `base`, G and mt are wave-uniform: the tile base stays in SGPRs and the load uses the saddr + 32-bit lane offset +
// immediate form (64-bit per-lane addresses cost two VGPRs per group and, under the register pressure of this kernel,
// were being spilled to scratch and reloaded one by one).
// Kernel experiments for the next round: the weight stream never hits in the 32-KiB vector L1 (a layer is
// 128 KiB), so it can be told not to allocate there -- -DLAB4D_A_NT: nt bit (streaming); -DLAB4D_A_SC: sc0 sc1 (system-coherent, L1 bypass).
__device__ __forceinline__ uint4 load_a(const GLOBAL_AS void* base, int G, int mt, int g, int lane) {
  const GLOBAL_AS char* tile = (const GLOBAL_AS char*)base + (size_t)(unsigned)(mt * G) * 1024u;
#if defined(LAB4D_A_NT)
  const u32x4_t v = __builtin_nontemporal_load((const GLOBAL_AS u32x4_t*)(tile + (unsigned)(g * 1024 + lane * 16)));
  return make_uint4(v.x, v.y, v.z, v.w);
#elif defined(LAB4D_A_SC)
  const u32x4_t v = *(const volatile GLOBAL_AS u32x4_t*)(tile + (unsigned)(g * 1024 + lane * 16));
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return gld16(tile + (unsigned)(g * 1024 + lane * 16));
#endif
}

// Kernel arguments are read from the kernarg segment at the point of use.  Left to itself the compiler hoists the ~60
// pointer loads of the argument block to the kernel entry, runs out of SGPRs and parks them in scratch (94 scratch
// stores at entry, reloads on every layer's critical path).  The empty asm makes the segment pointer opaque per use.
template <class T>
__device__ __forceinline__ GLOBAL_AS typename std::remove_pointer<T>::type* karg(size_t byte_off) {
  static_assert(std::is_pointer<T>::value, "karg: pointer arguments only");
  typedef __attribute__((address_space(4))) const char* kptr_t;  // constant address space: scalar loads
  kptr_t p = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  T v = *(__attribute__((address_space(4))) const T*)(p + byte_off);
  // a pointer loaded from memory is a generic (flat) pointer to the compiler; the result type carries the global address
  // space explicitly, otherwise every access through it becomes a flat_load/flat_store with 64-bit per-lane addressing
  return (GLOBAL_AS typename std::remove_pointer<T>::type*)v;
}
#define KARG_PTR(Struct, T, field, idx) karg<T>(offsetof(Struct, field) + sizeof(void*) * (size_t)(idx))

// acc += A(group) * B(unit)
template <class P>
__device__ __forceinline__ void mma_unit(f32x16_t& acc, const uint4& a, const uint4& b) {
  if constexpr (P::BF16) {
    bf16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
}

// accumulator tile -> the UPT B units of that 32-feature tile
template <class P>
__device__ __forceinline__ void tile_to_units(const f32x16_t& c, uint4* u) {
  if constexpr (P::BF16) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
      u[q] = make_uint4(pack2bf(c[8 * q + 0], c[8 * q + 1]), pack2bf(c[8 * q + 2], c[8 * q + 3]),
                        pack2bf(c[8 * q + 4], c[8 * q + 5]), pack2bf(c[8 * q + 6], c[8 * q + 7]));
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      u[e] = make_uint4(__float_as_uint(c[4 * e + 0]), __float_as_uint(c[4 * e + 1]), __float_as_uint(c[4 * e + 2]),
                        __float_as_uint(c[4 * e + 3]));
  }
}

// ---- [feature][sample] tile IO in accumulator layout -------------------------------------------
// A lane holds, per 4-feature group i, one dword per feature (bf16: samples (s0+2n, s0+2n+1) packed; fp32: sample
// s0+n).  Storing those dwords one by one costs 16 four-byte store instructions per tile and is TA-issue-bound
// (measured: the forward chain ran 2.3x slower with stores than without).  Instead the 4 lanes of a quad
// (n = 4k+q) transpose their 4x4 dword block with two butterfly exchanges, after which lane q owns ONE feature
// and 4 consecutive dwords (8 bf16 / 4 fp32 consecutive samples): one 16-byte store per group, 128-byte segments
// per feature row.  Loads are the mirror image.
__device__ __forceinline__ unsigned int quad_xor1(unsigned int v) { return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); }  // quad_perm [1,0,3,2]
__device__ __forceinline__ unsigned int quad_xor2(unsigned int v) { return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true); }  // quad_perm [2,3,0,1]
__device__ __forceinline__ void quad_transpose(unsigned int (&a)[4], int q) {
  // written so that every select takes ONE DPP-permuted operand straight from a register: the DPP move then folds into
  // the v_cndmask (v_cndmask_b32_dpp) and a 4x4 transpose costs 8 VALU ops instead of 16
  {
    const bool o = q & 1;
    const unsigned int x0 = quad_xor1(a[0]), x1 = quad_xor1(a[1]), x2 = quad_xor1(a[2]), x3 = quad_xor1(a[3]);
    a[0] = o ? x1 : a[0]; a[1] = o ? a[1] : x0; a[2] = o ? x3 : a[2]; a[3] = o ? a[3] : x2;
  }
  {
    const bool o = q & 2;
    const unsigned int x0 = quad_xor2(a[0]), x1 = quad_xor2(a[1]), x2 = quad_xor2(a[2]), x3 = quad_xor2(a[3]);
    a[0] = o ? x2 : a[0]; a[2] = o ? a[2] : x0; a[1] = o ? x3 : a[1]; a[3] = o ? a[3] : x1;
  }
}

// elements between consecutive 64-sample blocks: F*64 plus a 128-element skew.  Without the skew all concurrently
// running waves (which march through their tiles in near lock-step) hit addresses that are equal modulo the block
// size, i.e. the same HBM channel ("partition camping": measured 1.6x slower dgrad chain).
__host__ __device__ __forceinline__ constexpr size_t block_stride(int F) { return (size_t)F * 64 + 128; }

template <class P>
__device__ __forceinline__ size_t tile_byte_offset(int F, int s0, int row, int k) {
  return ((size_t)(s0 >> 6) * block_stride(F) + (size_t)row * 64 + (s0 & 63)) * sizeof(typename P::store_t) + 16 * (size_t)k;
}
// the same offset split for saddr addressing: wave-uniform part (tile + first row of the 32-row M-tile) ...
template <class P>
__device__ __forceinline__ size_t tile_base_offset(int F, int s0, int row0) {
  return ((size_t)(s0 >> 6) * block_stride(F) + (size_t)row0 * 64 + (s0 & 63)) * sizeof(typename P::store_t);
}
// ... and the per-lane part (row within the tile, 16-byte column k): < 32*64*4 + 128 bytes
template <class P>
__device__ __forceinline__ unsigned tile_lane_offset(int row, int k) {
  return (unsigned)(row * 64) * (unsigned)sizeof(typename P::store_t) + 16u * (unsigned)k;
}

// ---- packed-bf16 epilogue helpers (bf16 path) ------------------------------------------------------------------------------------
// The epilogue of an M-tile used to cost ~270 VALU instructions per 32 MFMAs (ReLU in fp32, sign bits by compare + select + or
// per value, two fp32->bf16 packings, 4x4 quad transposes as separate DPP moves and selects) -- above the ~5 issue slots per
// 32x32x16 MFMA a single wave per SIMD can hide (MI355X_MICROARCH.md).  Working on the PACKED pre-activation instead:
//   w[t][k] = bf16x2(acc[t][2k], acc[t][2k+1])   one v_cvt_pk_bf16_f32 per pair, exactly the next layer's B-operand dword
//   sign bits: bits = (bits >> 1) | (w & 0x80008000) per dword (v_lshrrev + v_and_or): 2 ops per PAIR
//   ReLU: v_pk_max_i16(w, 0) -- a bf16 with the sign bit set is a negative int16 -- relu(round(x)) == round(relu(x))
//   store dwords (two adjacent samples of one feature): one v_perm_b32 of the two n-tiles' packed words
//   quad transposes: v_cndmask_b32_dpp (the DPP operand folded into the select), 32 per tile, in one asm block
// the same conversion as an opaque instruction: with the builtin form the optimiser splits the pair again as soon as a bit
// operation touches one half of it (it then converts every value on its own and re-joins them with v_perm: 3 ops per pair)
__device__ __forceinline__ unsigned int pack2bf_op(float lo, float hi) {
  unsigned int w;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(lo), "v"(hi));
  return w;
}
typedef short i16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pk_relu_bf16(unsigned int w) {
  const i16x2_t v = __builtin_bit_cast(i16x2_t, w), z = {0, 0};
  return __builtin_bit_cast(unsigned int, __builtin_elementwise_max(v, z));
}
// halves of w multiplied by the 0/1 halves of m01 (exact masking of a packed bf16 pair in one v_pk_mul_lo_u16)
__device__ __forceinline__ unsigned int pk_mask_bf16(unsigned int w, unsigned int m01) {
  return __builtin_bit_cast(unsigned int, __builtin_bit_cast(u16x2_t, w) * __builtin_bit_cast(u16x2_t, m01));
}
// ReLU sign mask of one tile in the packed layout: bit (8t + k + 16*odd) belongs to accumulator register r = 2k + odd of n-tile t;
// 1 = the unit is ALIVE (pre-activation not negative).  w[t][k]: packed PRE-activation pairs.
__device__ __forceinline__ unsigned int pk_alive_bits(const unsigned int (&w)[2][8]) {
  unsigned int sb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    sb[t] = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) sb[t] = (sb[t] >> 1) | (w[t][k] & 0x80008000u);  // word k: low sign -> bit 8+k, high sign -> bit 24+k
  }
  return ~((sb[0] >> 8) | sb[1]);
}
__host__ __device__ __forceinline__ constexpr int pk_bit(int t, int r) { return 8 * t + (r >> 1) + 16 * (r & 1); }
// position of (n-tile t, register r) in a stored ReLU mask word: packed layout for bf16 tiles, plain 16t + r for fp32 tiles
template <class P>
__host__ __device__ __forceinline__ constexpr int mask_bit(int t, int r) { return P::BF16 ? pk_bit(t, r) : 16 * t + r; }
// 0/1 halves for packed word k of n-tile t from the alive bits
__device__ __forceinline__ unsigned int pk_m01(unsigned int alive, int t, int k) { return (alive >> (8 * t + k)) & 0x00010001u; }

// Four 4x4 dword transposes (one per 4-feature group of a 32-row tile) inside every lane quad: d[i][j], lane q of the quad ends up
// with the dwords j = 0..3 that lanes 0..3 held at index q.  32 v_cndmask_b32_dpp + 4 s_mov_b64 in ONE asm block: VOP2
// v_cndmask takes its condition from VCC only, so the four lane masks are switched in per phase, and the eight selects of a phase
// are independent (a DPP read needs two wait states after the VALU write of its source: phases keep writers >= 7 slots away).
// Results: c[i][0] = d[i][0], c[i][1] = d[i][2], c[i][2] = t2[i], c[i][3] = d[i][3] (returned through `out`).
__device__ __forceinline__ void quad_transpose4(unsigned int (&d)[4][4], unsigned int (&out)[4][4]) {
  unsigned int t0[4], t2[4];
  const unsigned long long mE = 0x5555555555555555ull, mO = 0xAAAAAAAAAAAAAAAAull, mL = 0x3333333333333333ull, mH = 0xCCCCCCCCCCCCCCCCull;
#define LAB4D_QP1 " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define LAB4D_QP2 " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
  // Four separate statements (the scheduler may put MFMAs between the phases; volatile keeps their order).  Every phase sets VCC
  // itself.  A DPP read needs two wait states after the VALU write of its source: phase A reads registers the code in front
  // of it has just written (s_mov + s_nop = 2 states), the later phases only read what an earlier phase wrote >= 8 slots back.
  // phase A: t0 = even ? d0 : xor1(d1) ; t2 = even ? d2 : xor1(d3)
  asm volatile("s_mov_b64 vcc, %[m]\n\ts_nop 0\n\t"
               "v_cndmask_b32_dpp %[t00], %[d01], %[d00], vcc" LAB4D_QP1 "v_cndmask_b32_dpp %[t20], %[d03], %[d02], vcc" LAB4D_QP1
               "v_cndmask_b32_dpp %[t01], %[d11], %[d10], vcc" LAB4D_QP1 "v_cndmask_b32_dpp %[t21], %[d13], %[d12], vcc" LAB4D_QP1
               "v_cndmask_b32_dpp %[t02], %[d21], %[d20], vcc" LAB4D_QP1 "v_cndmask_b32_dpp %[t22], %[d23], %[d22], vcc" LAB4D_QP1
               "v_cndmask_b32_dpp %[t03], %[d31], %[d30], vcc" LAB4D_QP1 "v_cndmask_b32_dpp %[t23], %[d33], %[d32], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
               : [t00] "=&v"(t0[0]), [t01] "=&v"(t0[1]), [t02] "=&v"(t0[2]), [t03] "=&v"(t0[3]), [t20] "=&v"(t2[0]), [t21] "=&v"(t2[1]),
                 [t22] "=&v"(t2[2]), [t23] "=&v"(t2[3])
               : [m] "s"(mE), [d00] "v"(d[0][0]), [d01] "v"(d[0][1]), [d02] "v"(d[0][2]), [d03] "v"(d[0][3]), [d10] "v"(d[1][0]), [d11] "v"(d[1][1]),
                 [d12] "v"(d[1][2]), [d13] "v"(d[1][3]), [d20] "v"(d[2][0]), [d21] "v"(d[2][1]), [d22] "v"(d[2][2]), [d23] "v"(d[2][3]),
                 [d30] "v"(d[3][0]), [d31] "v"(d[3][1]), [d32] "v"(d[3][2]), [d33] "v"(d[3][3])
               : "vcc");
  // phase B: d1 = odd ? d1 : xor1(d0) ; d3 = odd ? d3 : xor1(d2)
  asm volatile("s_mov_b64 vcc, %[m]\n\t"
               "v_cndmask_b32_dpp %[d01], %[d00], %[d01], vcc" LAB4D_QP1 "v_cndmask_b32_dpp %[d03], %[d02], %[d03], vcc" LAB4D_QP1
               "v_cndmask_b32_dpp %[d11], %[d10], %[d11], vcc" LAB4D_QP1 "v_cndmask_b32_dpp %[d13], %[d12], %[d13], vcc" LAB4D_QP1
               "v_cndmask_b32_dpp %[d21], %[d20], %[d21], vcc" LAB4D_QP1 "v_cndmask_b32_dpp %[d23], %[d22], %[d23], vcc" LAB4D_QP1
               "v_cndmask_b32_dpp %[d31], %[d30], %[d31], vcc" LAB4D_QP1 "v_cndmask_b32_dpp %[d33], %[d32], %[d33], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
               : [d01] "+v"(d[0][1]), [d03] "+v"(d[0][3]), [d11] "+v"(d[1][1]), [d13] "+v"(d[1][3]), [d21] "+v"(d[2][1]), [d23] "+v"(d[2][3]),
                 [d31] "+v"(d[3][1]), [d33] "+v"(d[3][3])
               : [m] "s"(mO), [d00] "v"(d[0][0]), [d02] "v"(d[0][2]), [d10] "v"(d[1][0]), [d12] "v"(d[1][2]), [d20] "v"(d[2][0]), [d22] "v"(d[2][2]),
                 [d30] "v"(d[3][0]), [d32] "v"(d[3][2])
               : "vcc");
  // phase C: c0 = low ? t0 : xor2(t2) ; c1 = low ? d1 : xor2(d3)
  unsigned int c0[4], c1[4];
  asm volatile("s_mov_b64 vcc, %[m]\n\t"
               "v_cndmask_b32_dpp %[c00], %[t20], %[t00], vcc" LAB4D_QP2 "v_cndmask_b32_dpp %[c10], %[d03], %[d01], vcc" LAB4D_QP2
               "v_cndmask_b32_dpp %[c01], %[t21], %[t01], vcc" LAB4D_QP2 "v_cndmask_b32_dpp %[c11], %[d13], %[d11], vcc" LAB4D_QP2
               "v_cndmask_b32_dpp %[c02], %[t22], %[t02], vcc" LAB4D_QP2 "v_cndmask_b32_dpp %[c12], %[d23], %[d21], vcc" LAB4D_QP2
               "v_cndmask_b32_dpp %[c03], %[t23], %[t03], vcc" LAB4D_QP2 "v_cndmask_b32_dpp %[c13], %[d33], %[d31], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
               : [c00] "=&v"(c0[0]), [c01] "=&v"(c0[1]), [c02] "=&v"(c0[2]), [c03] "=&v"(c0[3]), [c10] "=&v"(c1[0]), [c11] "=&v"(c1[1]),
                 [c12] "=&v"(c1[2]), [c13] "=&v"(c1[3])
               : [m] "s"(mL), [t00] "v"(t0[0]), [t01] "v"(t0[1]), [t02] "v"(t0[2]), [t03] "v"(t0[3]), [t20] "v"(t2[0]), [t21] "v"(t2[1]), [t22] "v"(t2[2]),
                 [t23] "v"(t2[3]), [d01] "v"(d[0][1]), [d03] "v"(d[0][3]), [d11] "v"(d[1][1]), [d13] "v"(d[1][3]), [d21] "v"(d[2][1]), [d23] "v"(d[2][3]),
                 [d31] "v"(d[3][1]), [d33] "v"(d[3][3])
               : "vcc");
  // phase D: c2 = high ? t2 : xor2(t0) ; c3 = high ? d3 : xor2(d1)
  asm volatile("s_mov_b64 vcc, %[m]\n\t"
               "v_cndmask_b32_dpp %[t20], %[t00], %[t20], vcc" LAB4D_QP2 "v_cndmask_b32_dpp %[d03], %[d01], %[d03], vcc" LAB4D_QP2
               "v_cndmask_b32_dpp %[t21], %[t01], %[t21], vcc" LAB4D_QP2 "v_cndmask_b32_dpp %[d13], %[d11], %[d13], vcc" LAB4D_QP2
               "v_cndmask_b32_dpp %[t22], %[t02], %[t22], vcc" LAB4D_QP2 "v_cndmask_b32_dpp %[d23], %[d21], %[d23], vcc" LAB4D_QP2
               "v_cndmask_b32_dpp %[t23], %[t03], %[t23], vcc" LAB4D_QP2 "v_cndmask_b32_dpp %[d33], %[d31], %[d33], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
               : [t20] "+v"(t2[0]), [t21] "+v"(t2[1]), [t22] "+v"(t2[2]), [t23] "+v"(t2[3]), [d03] "+v"(d[0][3]), [d13] "+v"(d[1][3]), [d23] "+v"(d[2][3]),
                 [d33] "+v"(d[3][3])
               : [m] "s"(mH), [t00] "v"(t0[0]), [t01] "v"(t0[1]), [t02] "v"(t0[2]), [t03] "v"(t0[3]), [d01] "v"(d[0][1]), [d11] "v"(d[1][1]), [d21] "v"(d[2][1]),
                 [d31] "v"(d[3][1])
               : "vcc");
#undef LAB4D_QP1
#undef LAB4D_QP2
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[i][0] = c0[i];
    out[i][1] = c1[i];
    out[i][2] = t2[i];
    out[i][3] = d[i][3];
  }
}

// store one 32-row tile from the packed words of both n-tiles: w[t][k] = bf16x2(value of register 2k, value of register 2k+1)
__device__ __forceinline__ void store_tile_packed(GLOBAL_AS void* buf, int F, int s0, int mt, int lane, const unsigned int (&w)[2][8]) {
#ifdef LAB4D_ABL_L2STORE  // kernel experiment (timing only): every wave keeps writing the same 2048-sample window, so the stores never reach HBM
  s0 &= 0x7ff;
#endif
  const int n = lane & 31, h = lane >> 5, q = n & 3, k = n >> 2;
  GLOBAL_AS char* base = (GLOBAL_AS char*)buf + tile_base_offset<PBF16>(F, s0, 32 * mt);
  const unsigned lo = tile_lane_offset<PBF16>(4 * h + q, k);
  unsigned int d[4][4], c[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // register r = 4i + j: dword = (sample 2n of n-tile 0, sample 2n+1 of n-tile 1) of that feature
      const int r = 4 * i + j;
      d[i][j] = __builtin_amdgcn_perm(w[1][r >> 1], w[0][r >> 1], (r & 1) ? 0x07060302u : 0x05040100u);
    }
  quad_transpose4(d, c);
#pragma unroll
  for (int i = 0; i < 4; ++i) gst16o(base, lo + tile_lane_offset<PBF16>(8 * i, 0), c[i][0], c[i][1], c[i][2], c[i][3]);
}


// ---- tile store through the LDS transpose-read (round 2, second session) ----------------------------------------------------------
// The packed units of a finished tile sit in the wave's slab anyway ([n-tile][unit][lane] 16 B: 8 features of one sample).  gfx950's
// ds_read_b64_tr_b16 is a free 4x4 16-bit transpose with ARBITRARY per-lane chunk addresses (tools/probes/tr_probe.hip, verified on
// hardware: inside a 16-lane group, lane i's element j is element i&3 of the 8-byte chunk addressed by lane (i>>2) + 4j).  With lane
// c = (m = c&3, j = c>>2) of group G = (h = G&1, S = G>>1) pointing at sample 32S + 8m + j (and + 4 for a second read), lane i
// receives samples 32S + 8(i>>2) + 0..7 of ONE feature = exactly a 16-byte piece of the [feature][64 samples] row: 8 reads + 4 stores per
// 32 x 64 tile and NO VALU work (the DPP network it replaces: 16 v_perm + 32 v_cndmask_dpp per tile).  The reads are issued right
// after the slab writes and waited for in `flush`, behind the next tile's weight requests.
// Measured on the shipped one-wave-per-SIMD kernels (-DLAB4D_TRSTORE): parity-green, forward unchanged (6.71 vs 6.74 ms), backward
// slower (8.42 vs 7.91 ms): with nothing to switch to, 8 more LDS round trips per step cost more than 48 VALU slots save -- so the
// DPP network stays the default HERE.  The path is kept because it is what makes 32-sample tiles (two waves per SIMD for the
// 256-wide nets) practical: a 32 x 32 tile needs an 8 x 8 16-bit transpose in registers, but only 4 of these reads.
struct TrTile {
  unsigned long long a[4], b[4];  // [2q + a]: first / second read of a pair
};
__device__ __forceinline__ unsigned tr_lane_base(unsigned slab_lds, int UW, int lane) {
  // LDS byte address of this lane's supplier chunk for (mt = 0, q = 0, a = 0, first read)
  const int i = lane & 15, G = lane >> 4, h = G & 1, S = G >> 1, m = i & 3, j = i >> 2;
  const int sigma = 32 * S + 8 * m + j, n = sigma >> 1, t = sigma & 1;
  return slab_lds + (unsigned)(((t * UW) * 64 + 32 * h + n) * 16);
}
__device__ __forceinline__ void tr_issue(unsigned addr /* tr_lane_base + mt * 2048 */, TrTile& r) {
  // second read of a pair: samples + 4 = lane n + 2 = + 32 bytes; q: next unit = + 1024 bytes; a: + 8 bytes
  asm volatile("ds_read_b64_tr_b16 %0, %8\n\t"
               "ds_read_b64_tr_b16 %4, %8 offset:32\n\t"
               "ds_read_b64_tr_b16 %1, %8 offset:8\n\t"
               "ds_read_b64_tr_b16 %5, %8 offset:40\n\t"
               "ds_read_b64_tr_b16 %2, %8 offset:1024\n\t"
               "ds_read_b64_tr_b16 %6, %8 offset:1056\n\t"
               "ds_read_b64_tr_b16 %3, %8 offset:1032\n\t"
               "ds_read_b64_tr_b16 %7, %8 offset:1064"
               : "=&v"(r.a[0]), "=&v"(r.a[1]), "=&v"(r.a[2]), "=&v"(r.a[3]), "=&v"(r.b[0]), "=&v"(r.b[1]), "=&v"(r.b[2]), "=&v"(r.b[3])
               : "v"(addr)
               : "memory");
}
__device__ __forceinline__ void tr_wait(TrTile& r) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.a[2]), "+v"(r.a[3]), "+v"(r.b[0]), "+v"(r.b[1]), "+v"(r.b[2]), "+v"(r.b[3]));
}
__device__ __forceinline__ void tr_store(GLOBAL_AS void* buf, int F, int s0, int mt, int lane, const TrTile& r) {
#ifdef LAB4D_ABL_L2STORE
  s0 &= 0x7ff;
#endif
  const int i = lane & 15, G = lane >> 4, h = G & 1, S = G >> 1;
  GLOBAL_AS char* base = (GLOBAL_AS char*)buf + tile_base_offset<PBF16>(F, s0, 32 * mt);
  const unsigned lo = (unsigned)((4 * h + (i & 3)) * 128 + (32 * S + 8 * (i >> 2)) * 2);
#pragma unroll
  for (int qa = 0; qa < 4; ++qa) {  // qa = 2q + a: rows 16q + 8a + 4h + (i&3)
    const unsigned long long A = r.a[qa], B = r.b[qa];
    gst16(base + (lo + (unsigned)((16 * (qa >> 1) + 8 * (qa & 1)) * 128)), (unsigned)A, (unsigned)(A >> 32), (unsigned)B, (unsigned)(B >> 32));
  }
}

// The same tile IO in four independent pieces (qa = 2q + a: rows 16q + 8a + ..): two transposing reads, one 16-byte store each.
// -DLAB4D_TRSPREAD issues the pieces BETWEEN the MFMAs of the second half of the step the tile was finished in, instead of four
// back-to-back stores per wave at the end of every step (the counters say the vector L1 stalls the store data path for a quarter
// of the kernel: profiles/r02_stall_counters.txt).  The data waits in the wave's slab, not in registers.
struct TrPiece {
  unsigned long long a, b;
};
template <int QA>
__device__ __forceinline__ void trp_issue(unsigned addr /* tr_lane_base + mt * 2048 */, TrPiece& r) {
  constexpr int OFF = 1024 * (QA >> 1) + 8 * (QA & 1);
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\t"
               "ds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(r.a), "=&v"(r.b)
               : "v"(addr), "n"(OFF), "n"(OFF + 32)
               : "memory");
}
template <int QA>
__device__ __forceinline__ void trp_store(GLOBAL_AS void* buf, int F, int s0, int mt, int lane, TrPiece& r) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.a), "+v"(r.b));
  const int i = lane & 15, G = lane >> 4, h = G & 1, S = G >> 1;
  GLOBAL_AS char* base = (GLOBAL_AS char*)buf + tile_base_offset<PBF16>(F, s0, 32 * mt);
  const unsigned lo = (unsigned)((4 * h + (i & 3)) * 128 + (32 * S + 8 * (i >> 2)) * 2);
  gst16(base + (lo + (unsigned)((16 * (QA >> 1) + 8 * (QA & 1)) * 128)), (unsigned)r.a, (unsigned)(r.a >> 32), (unsigned)r.b, (unsigned)(r.b >> 32));
}

template <class P>
__device__ __forceinline__ void store_tile(GLOBAL_AS void* buf, int F, int s0, int mt, int lane, const f32x16_t* c /*[NT]*/) {
  const int n = lane & 31, h = lane >> 5, q = n & 3, k = n >> 2;
  GLOBAL_AS char* base = (GLOBAL_AS char*)buf + tile_base_offset<P>(F, s0, 32 * mt);
  const unsigned lo = tile_lane_offset<P>(4 * h + q, k);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned int d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (P::BF16) d[j] = pack2bf(c[0][4 * i + j], c[1][4 * i + j]);
      else d[j] = __float_as_uint(c[0][4 * i + j]);
    }
    quad_transpose(d, q);
    gst16(base + (lo + tile_lane_offset<P>(8 * i, 0)), d[0], d[1], d[2], d[3]);
  }
}
// load_tile in two halves so that the four 16-byte loads can be issued one pipeline step ahead of their use
template <class P>
__device__ __forceinline__ void load_tile_raw(const GLOBAL_AS void* buf, int F, int s0, int mt, int lane, uint4 (&raw)[4]) {
  const int n = lane & 31, h = lane >> 5, q = n & 3, k = n >> 2;
  const GLOBAL_AS char* base = (const GLOBAL_AS char*)buf + tile_base_offset<P>(F, s0, 32 * mt);
  const unsigned lo = tile_lane_offset<P>(4 * h + q, k);
#pragma unroll
  for (int i = 0; i < 4; ++i) raw[i] = gld16(base + (lo + tile_lane_offset<P>(8 * i, 0)));
}
template <class P>
__device__ __forceinline__ void tile_from_raw(const uint4 (&raw)[4], int lane, f32x16_t* c /*[NT]*/) {
  const int q = lane & 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned int d[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
    quad_transpose(d, q);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (P::BF16) {
        c[0][4 * i + j] = bf2f((unsigned short)(d[j] & 0xffffu));
        c[1][4 * i + j] = bf2f((unsigned short)(d[j] >> 16));
      } else {
        c[0][4 * i + j] = __uint_as_float(d[j]);
      }
    }
  }
}
template <class P>
__device__ __forceinline__ void load_tile(const GLOBAL_AS void* buf, int F, int s0, int mt, int lane, f32x16_t* c /*[NT]*/) {
  uint4 raw[4];
  load_tile_raw<P>(buf, F, s0, mt, lane, raw);
  tile_from_raw<P>(raw, lane, c);
}

// ---- kernel argument blocks (passed by value) ---------------------------------------------------
struct FwdK {
  int S, S_pad, ld, spf, ntiles;
  const float* x;
  const float* freq_w;
  const void* W[LAB4D_MLP_MAX_LAYERS];
  const float* bias[LAB4D_MLP_MAX_LAYERS];
  const float* pf_bias[LAB4D_MLP_MAX_LAYERS];
  void* act[LAB4D_MLP_MAX_LAYERS];
  unsigned int* mask[LAB4D_MLP_MAX_LAYERS];  // ReLU sign bits, [tile][m_tile][lane] (bit 16t+r), or NULL
  void* emb;
  const void* ext;
  float* out;
  const float* x2;  // nets with AUX3: second per-sample 3-vector (view direction), raw embedding slots 6L+3..6L+5
  const int* S_dev;      // device-side sample count (compacted evaluation: the count never visits the host), or NULL
  const int* frame_idx;  // (S) frame of every sample (compacted samples are not frame-contiguous), or NULL: frame = s / spf
  const float* aff;      // raw-input nets: (M, CIN, 4) per-frame affine rows -- x is then the (S,3) points and the inputs are formed here
};
struct BwdK {
  int S, S_pad, ld, spf, ntiles;
  const void* WT[LAB4D_MLP_MAX_LAYERS];
  const void* act[LAB4D_MLP_MAX_LAYERS];
  const unsigned int* mask[LAB4D_MLP_MAX_LAYERS];
  const void* emb;
  const void* ext;
  const float* d_out;
  const void* ext_gin;
  void* ext_gout;
  void* dz[LAB4D_MLP_MAX_LAYERS];
  float* d_x;
  float* d_x2;  // nets with AUX3: gradient wrt the second 3-vector, or NULL
  const float* x;    // EMB == 2 nets: the (S,3) points, the (M, KE, 4) per-frame table of the affine first layer, and its gradient (accumulated)
  const float* aff;
  float* g_aff;
};

// values of embedding slots (2*pair, 2*pair+1) for a point x -- posenc nets
template <class Net>
__device__ __forceinline__ void emb_pair(int pair, const float* x /* [6]: point, then the aux 3-vector (or zeros) */, const float* freq_w, float& v0, float& v1) {
  // pair p < 3L: (sin, cos)(2^f x_a) * w_f with f = p / 3, a = p % 3 ; then x_0,x_1,x_2 ; then zeros
  constexpr int L = Net::NFREQ;
  if (pair < 3 * L) {
    const int f = pair / 3, a = pair - 3 * f;
    const float xa = a == 0 ? x[0] : (a == 1 ? x[1] : x[2]);
    const float ang = ldexpf(xa, f);  // exact 2^f * x  (embedding.py:50,104)
    float s, c;
    sincosf(ang, &s, &c);
    const float w = freq_w ? freq_w[f] : 1.0f;
    v0 = s * w;
    v1 = c * w;
  } else {
    const int s0 = 2 * pair - 6 * L;  // slot index relative to the raw block: [x (3) | aux (3, nets with AUX3) | zeros]
    constexpr int NR = Net::AUX3 ? 6 : 3;
    v0 = s0 < NR ? x[s0 < NR ? s0 : 0] : 0.f;
    v1 = s0 + 1 < NR ? x[s0 + 1 < NR ? s0 + 1 : 0] : 0.f;
  }
}

// ---- workgroup-shared weight stream ---------------------------------------------------------------------------
// The four waves of a workgroup walk the layers and M-tiles in lock-step (same code, same trip counts), so the 16 KiB of
// packed weights one M-tile step needs is the SAME for all of them.  Fetched per wave it was 4 x 16 KiB through the
// CU's vector-memory path per step (measured: removing those loads made the training-mode forward 1.57x faster).
// Instead every wave fetches a quarter of the NEXT step's A groups into registers, drops them into an LDS buffer at
// the end of the step, and after ONE barrier per step all waves read the groups they need back (lane-linear
// ds_read_b128, conflict-free).  Up to ACACHE_G groups go through LDS (2 buffers x 16 KiB: with the four 32 KiB slabs
// that is the CU's whole 160 KiB); the few groups beyond (skip-layer embedding columns) keep the direct path.
#ifndef LAB4D_ACACHE_G
#define LAB4D_ACACHE_G 14
#endif
constexpr int ACACHE_G = LAB4D_ACACHE_G;  // 2 x 14 KiB: leaves 4 KiB of the 160 KiB unallocated (a kernel that needs ALL of the LDS cannot be co-scheduled with anything, e.g. a profiler's helper)
// The narrow fg nets (feature 128 wide, visibility 64 wide; bf16) are built for TWO workgroups per CU (two waves per SIMD from independent, not lock-stepped
// workgroups: while one waits at its barrier / on a store the other issues MFMAs): <= 256 registers per lane
// (__launch_bounds__(256, 2)) and <= 78 KiB of LDS (4 x 16 KiB slabs + 2 x 7 KiB of shared A groups).
template <class Net, class P, bool BWD = false>
constexpr int want_occ() {
#ifdef LAB4D_ABL_OCC1
  return 1;
#else
  // (the affine-form skin nets were tried at 2: their forward spills 126-158 registers at 256, their backward carries 128 accumulators of the table gradient)
  return (P::BF16 && (Net::ID == LAB4D_NET_FEAT || Net::ID == LAB4D_NET_VIS)) ? 2 : 1;  // the bg nets (per-frame bias in two layers) spill 25-53 registers at 256
#endif
}
template <class Net, class P>
constexpr int acache_g() {
#ifdef LAB4D_ABL_ACG14
  return ACACHE_G;
#elif defined(LAB4D_ABL_ACG7)
  return (P::BF16 && (Net::ID == LAB4D_NET_FEAT || Net::ID == LAB4D_NET_VIS)) ? 7 : ACACHE_G;
#else
  return want_occ<Net, P>() == 2 ? 7 : ACACHE_G;
#endif
}
__device__ __forceinline__ void wg_step_barrier() {
  // LDS writes of this wave visible + everybody arrived.  Raw s_barrier: __syncthreads() would also drain the
  // outstanding activation stores (vmcnt(0)), a full HBM round trip per step.
#ifdef LAB4D_ABL_NOBAR  // kernel experiment (timing only, results wrong): no lock-step
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// ---- LDS-DMA weight stream (round 4, -DLAB4D_ADMA; backward chain, bf16) -------------------------------------------------------------------
// The shared A groups of a step go global -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave-instruction, lane-linear, no staging registers,
// no ds_write), issued right behind the step barrier into the buffer the previous step finished reading -- one step of lead instead of two,
// 16 registers and 4 LDS stores per wave and step less.  Two things make it work with compiler-managed waits: (1) the DMA is the BUILTIN, so the
// waitcnt pass counts it like any other vector-memory operation (an inline-asm DMA is invisible to it and shifts every counted vmcnt of the step
// by the number of hidden pieces: each wait then also drains the previous step's tile stores); (2) the two buffers are two separate __shared__
// arrays selected by compile-time constants, so a read of the buffer that landed a step ago is not ordered behind the DMA that has just been
// issued into the other one (the pass tracks LDS-DMA destinations per LDS variable).  What the compiler cannot know is the cross-wave part: a
// wave's pieces must have landed before it arrives at the step barrier (wg_step_barrier_dma: all but the N most recent vector-memory operations --
// the tile stores issued at the end of the step -- are waited for).
__device__ __forceinline__ void a_dma_1k(const GLOBAL_AS void* gsrc_lane, __attribute__((address_space(3))) void* lds_base_uniform) {
#if defined(LAB4D_ADMA) && LAB4D_ADMA == 2  // experiment: the same transfer as inline asm (invisible to the waitcnt pass; M0 is used by nothing else in these kernels)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc_lane), "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)lds_base_uniform)) : "memory");
#else
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS unsigned int*)gsrc_lane, (__attribute__((address_space(3))) unsigned int*)lds_base_uniform, 16, 0, 0);
#endif
}
template <int KEEP>
__device__ __forceinline__ void wg_step_barrier_dma() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(KEEP) : "memory");
}
template <class P>
constexpr int bwd_step_stores() {  // the dZ tile stores flush_act issues at the very end of a backward step (store_tile_packed: four 16-byte stores)
#if defined(LAB4D_ABL_NOSTORE) || defined(LAB4D_TRSPREAD) || defined(LAB4D_TRSTORE)
  return 0;
#else
  return P::BF16 ? 4 : 0;
#endif
}
template <class P>
constexpr bool use_adma() {
#ifdef LAB4D_ADMA
  return P::BF16;
#else
  return false;
#endif
}

// One wave per SIMD issues in order: an MFMA that is followed in the instruction stream by a long run of VALU work leaves the matrix pipe
// idle for the whole run, and a run of back-to-back MFMAs leaves the VALU idle.  The machine scheduler interleaves the two only
// partly (tools/isa_blocks.py: runs of 40-100 vector instructions without an MFMA at the end of every pipeline step), so the step
// regions ask for the pattern explicitly: NM times {1 MFMA, NV VALU}.  Scheduling only -- results are unaffected.
#ifndef LAB4D_SCHED_NV
#define LAB4D_SCHED_NV 5
#endif
#ifndef LAB4D_SCHED_NV_FWD
#define LAB4D_SCHED_NV_FWD 5
#endif
#ifndef LAB4D_SCHED_FWD_ON  // forward chains: every pattern tried (3..7 VALU per MFMA) made the static schedule WORSE than the scheduler's own; off
#define LAB4D_SCHED_FWD_ON false
#endif
template <class Net>
constexpr bool sched_il_bwd() { return Net::ID != LAB4D_NET_FG_COLOR; }  // the colour net's backward spills 10 registers with the pattern
template <int NM, int NV, bool ON = true>
__device__ __forceinline__ void sched_interleave() {
#ifdef LAB4D_SCHED_IL
#pragma unroll
  for (int i = 0; i < (ON ? NM : 0); ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);  // VALU
  }
#endif
}

// ---- layer kinds ----------------------------------------------------------------------------------
// Layers of identical shape share ONE copy of the layer code, executed from a runtime loop over the layers: the fully
// unrolled chain (10 layer bodies, 70-120 KB of code) does not fit the 64 KB instruction cache two CUs share, and with
// every wave at a different layer the cache thrashed (measured: the training-mode forward ran 1.4x faster with the
// seven 256x256 layers sharing their code).  A kind is named by its first ("representative") layer.
template <class Net>
constexpr bool fwd_same(int a, int b) {
  const LS x = Net::L[a], y = Net::L[b];
  return x.ke == y.ke && x.kin == y.kin && x.mout == y.mout && x.relu == y.relu && x.pf == y.pf && x.add_ext == y.add_ext &&
         ((a == Net::NL - 1) == (b == Net::NL - 1));
}
template <class Net>
constexpr int fwd_rep(int l) {
  for (int j = 0; j < l; ++j)
    if (fwd_same<Net>(j, l)) return j;
  return l;
}
template <class Net>
constexpr unsigned fwd_members(int r) {  // bit l set: layer l runs the code of representative r
  unsigned m = 0;
  for (int l = 0; l < Net::NL; ++l)
    if (fwd_rep<Net>(l) == r) m |= 1u << l;
  return m;
}
template <class Net>
constexpr bool fwd_any_export(int r) {
  for (int l = 0; l < Net::NL; ++l)
    if (fwd_rep<Net>(l) == r && Net::L[l].ext_grad) return true;
  return false;
}
// backward: the body of layer l depends on its own shape and on the layer below (mask / ext of its output)
template <class Net>
constexpr bool bwd_same(int a, int b) {
  if ((a == 0) != (b == 0)) return false;
  const LS x = Net::L[a], y = Net::L[b];
  if (!(x.ke == y.ke && x.kin == y.kin && x.mout == y.mout)) return false;
  if (a == 0) return true;
  const LS xp = Net::L[a - 1], yp = Net::L[b - 1];
  return xp.relu == yp.relu && xp.ext_grad == yp.ext_grad && xp.add_ext == yp.add_ext && xp.mout == yp.mout;
}
template <class Net>
constexpr int bwd_rep(int l) {
  for (int j = Net::NL - 1; j > l; --j)
    if (bwd_same<Net>(j, l)) return j;
  return l;
}
template <class Net>
constexpr unsigned bwd_members(int r) {
  unsigned m = 0;
  for (int l = 0; l < Net::NL; ++l)
    if (bwd_rep<Net>(l) == r) m |= 1u << l;
  return m;
}

// heads with many channels (the delta-skin logits: 25 / 18) move their (TILE, COUT) fp32 tiles through the wave's slab (coalesced global IO).
// (The 16-channel feature head stays on per-lane rows: staged, its two-workgroups-per-CU kernels spill 6-7 more registers.)
template <class Net>
constexpr bool head_staged() { return Net::COUT > 16; }

template <class Net, class P>
struct Slab {
  static constexpr int W = net_wmax<Net>();
  static constexpr int UW = W / P::FPG;                 // units per n-tile
  // raw-input nets (and the tangent forward) stage a tile of their (S, C) fp32 input / input gradient through the slab:
  // TILE x C floats, copied to / from global memory as one contiguous, coalesced region
  static constexpr int STAGE_C = Net::EMB == 1 ? Net::CIN : Net::KE;
  static constexpr bool STAGES = Net::EMB == 1 || Net::ID == LAB4D_NET_FG_BASE || Net::ID == LAB4D_NET_BG_BASE;  // raw input, or the tangent-mode forward
  static constexpr int UNITS_STAGE = STAGES ? (P::TILE * STAGE_C * 4 + 15) / 16 : 0;
  static constexpr int UNITS_LAYER = P::NT * UW * 64;
  static constexpr int UNITS_PER_WAVE = UNITS_LAYER > UNITS_STAGE ? UNITS_LAYER : UNITS_STAGE;  // uint4 slots
};

// coalesced copy of a tile's TILE x C fp32 rows between global memory (rows s0.., clamped to the last valid element) and
// the wave-private staging area.  The (S, C) input of a raw-input net has a row stride of C*4 bytes (300 B for the 75
// bone coordinates): read per lane it costs one cache line per lane and load (measured: the 20 k-MAC skin net ran as
// long as a 160 k-MAC colour net).
template <int COUNT>
__device__ __forceinline__ void stage_in(float* __restrict__ stage, const float* __restrict__ x, long e0, long e_last, int lane) {
  const GLOBAL_AS float* gx = (const GLOBAL_AS float*)x;
  if (e0 + COUNT - 1 <= e_last && (((size_t)(x + e0)) & 15) == 0 && (COUNT & 3) == 0) {
    // all loads are issued before the first LDS write (a load -> wait -> write loop pays one memory latency per trip)
    constexpr int N4 = COUNT / 4, TRIPS = (N4 + 63) / 64;
    f32x4_t v[TRIPS];
#pragma unroll
    for (int i = 0; i < TRIPS; ++i) {
      const int e = lane + 64 * i;
      if (e < N4) v[i] = *(const GLOBAL_AS f32x4_t*)(gx + e0 + 4 * e);
    }
#pragma unroll
    for (int i = 0; i < TRIPS; ++i) {
      const int e = lane + 64 * i;
      if (e < N4) reinterpret_cast<float4*>(stage)[e] = make_float4(v[i].x, v[i].y, v[i].z, v[i].w);
    }
  } else {
    for (int e = lane; e < COUNT; e += 64) {
      const long ge = e0 + e;
      stage[e] = gx[ge <= e_last ? ge : e_last];
    }
  }
}
__device__ __forceinline__ void stage_out(const float* __restrict__ stage, float* __restrict__ y, long e0, int count, long e_last, int lane) {
  GLOBAL_AS float* gy = (GLOBAL_AS float*)y;
  if (e0 + count - 1 <= e_last && (((size_t)(y + e0)) & 15) == 0 && (count & 3) == 0) {
    for (int e = lane; e < count / 4; e += 64) {
      const float4 v = reinterpret_cast<const float4*>(stage)[e];
      f32x4_t o = {v.x, v.y, v.z, v.w};
      *(GLOBAL_AS f32x4_t*)(gy + e0 + 4 * e) = o;
    }
  } else {
    for (int e = lane; e < count; e += 64) {
      const long ge = e0 + e;
      if (ge <= e_last) gy[ge] = stage[e];
    }
  }
}

// =================================================================================================
// forward chain
// =================================================================================================
// TAN = tangent mode (eikonal term, nerf.py:416-453): the input is a raw (S, KE) tangent vector in embedding-slot order,
// layers have no bias, and ReLU is replaced by the sign bits the primal pass stored (the network is piecewise linear, so
// d/dtheta of the directional derivative of sdf is an ordinary backward pass of this masked linear network).
// ST = training mode: the embedding, every hidden post-activation and every ReLU sign mask are stored unconditionally
// (all pointers non-NULL, checked by the host).  It is a template parameter, not a runtime test, because a store inside
// a runtime branch makes the compiler's s_waitcnt vmcnt(N) bookkeeping assume the store-free path: the wait for the
// prefetched A groups of the next tile then also waits for this tile's activation stores (a full HBM round trip per tile).
// ACTS = false (with ST): only the embedding and the ReLU sign words are stored -- all a backward that wants nothing but d/dx needs (the eval path's
// normals, nerf.py:455-493: 38 GB of activations per 8.4 M samples neither written nor allocated)
template <class Net, class P, bool TAN = false, bool ST = false, bool ACTS = true>
__global__ void __launch_bounds__(256, (want_occ<Net, P>())) k_mlp_fwd(FwdK a) {
  constexpr int NT = P::NT, TILE = P::TILE, KE = Net::KE, UE = KE / P::FPG, UW = Slab<Net, P>::UW;
  constexpr int ACG = acache_g<Net, P>();
  __shared__ uint4 slab_all[4 * Slab<Net, P>::UNITS_PER_WAVE];
  __shared__ uint4 abuf[2 * ACG * 64];  // workgroup-shared A groups of the current / next M-tile step
  __shared__ float4 afftab[(Net::EMB != 0 && !TAN) ? 4 * 96 : 1];  // raw-input nets: the current frame's affine rows, one copy per wave
  const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
  // the wave index is the same in all lanes, but the compiler only knows that after a readfirstlane: with it the tile
  // index and every tile base address are scalar (SGPR) values instead of 64-bit per-lane VGPR pairs
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint4* slab = slab_all + wid * Slab<Net, P>::UNITS_PER_WAVE + lane;  // + (t*UW + u)*64
  const int wave = blockIdx.x * 4 + wid, nwaves = gridDim.x * 4;
  // LDS byte address of this lane's chunk for the transposing tile store (see tr_issue)
  const unsigned tr_base = tr_lane_base((unsigned)(size_t)(__attribute__((address_space(3))) uint4*)(slab_all + wid * Slab<Net, P>::UNITS_PER_WAVE), UW, lane);
  TrTile trt;

  // device-side sample count: only the first *S_dev samples exist (stream-compacted evaluation); tiles beyond them are skipped
  int S_eff = a.S, ntiles = a.ntiles;
  if (a.S_dev) {
    const int sd = *(const GLOBAL_AS int*)a.S_dev;
    S_eff = sd < a.S ? sd : a.S;
    const int nt = ((S_eff + TILE - 1) / TILE + 3) & ~3;
    ntiles = nt < a.ntiles ? nt : a.ntiles;
  }
  // ntiles is a multiple of 4 (host contract: S_pad % 256 == 0), so all four waves of a workgroup make the same number of
  // trips -- they meet at one barrier per M-tile step
  for (int tile = wave; tile < ntiles; tile += nwaves) {
    const int s0 = tile * TILE;
    int sidx[NT], frame[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int s = s0 + NT * n + t;
      sidx[t] = s;
      const int sc = s < S_eff ? s : S_eff - 1;  // padded tail recomputes the last sample (finite, never written out)
      frame[t] = a.frame_idx ? ((const GLOBAL_AS int*)a.frame_idx)[sc] : sc / a.spf;
    }
    // ---- embedding as B units (identity slot order) ----
    uint4 emb[NT][UE];
    constexpr bool RAW = (Net::EMB == 1) || TAN;
    constexpr int CINR = TAN ? KE : Net::CIN;
    float* stagef = reinterpret_cast<float*>(slab_all + wid * Slab<Net, P>::UNITS_PER_WAVE);
    if constexpr (RAW) {
      bool staged = false;
      if constexpr (Net::EMB == 1 && !TAN) {
        if (a.aff != nullptr) {
          // Fused bone coordinates: the tile's (TILE, CIN) input rows are FORMED here, c = aff[frame][c][0..2] . x + aff[frame][c][3],
          // and dropped into the staging area the copy below would have filled -- the (S, 3B) tensor (300 B per sample written by a
          // kernel of its own and read back here) never exists.  The two lane halves of a sample split its CIN columns.
          static_assert(Net::CIN <= 96, "affine table");
          constexpr int HALF = (Net::CIN + 1) / 2;
          const GLOBAL_AS float4* tabg = (const GLOBAL_AS float4*)a.aff;
          float4* ltab = afftab + wid * 96;
          const bool uni = (a.spf % TILE) == 0 && a.frame_idx == nullptr;  // the tile lies in one frame: stage its rows in LDS once
          if (uni) {
            const int m = __builtin_amdgcn_readfirstlane(frame[0]);
            for (int e = lane; e < Net::CIN; e += 64) {
              const f32x4_t r = *(const GLOBAL_AS f32x4_t*)(tabg + (size_t)m * Net::CIN + e);
              ltab[e] = make_float4(r.x, r.y, r.z, r.w);
            }
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int s = sidx[t] < S_eff ? sidx[t] : S_eff - 1;
            const float x0 = a.x[(size_t)s * 3], x1 = a.x[(size_t)s * 3 + 1], x2_ = a.x[(size_t)s * 3 + 2];
            float* row = stagef + (NT * n + t) * Net::CIN;
            for (int i = 0; i < HALF; ++i) {
              const int c = i + h * HALF;
              if (c < Net::CIN) {
                float4 r;
                if (uni) r = ltab[c];
                else {
                  const f32x4_t g = *(const GLOBAL_AS f32x4_t*)(tabg + (size_t)frame[t] * Net::CIN + c);
                  r = make_float4(g.x, g.y, g.z, g.w);
                }
                row[c] = r.x * x0 + r.y * x1 + r.z * x2_ + r.w;
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
          staged = true;
        }
      }
      if (!staged) stage_in<TILE * CINR>(stagef, a.x, (long)s0 * CINR, (long)S_eff * CINR - 1, lane);
    }
    if constexpr (Net::EMB == 2 && !TAN) {
      // Per-frame affine first layer (lab4d_mlp.h, LAB4D_NET_SKIN_A): slot j = relu(aff[frame][j] . [x; 1]), fp32 FMAs straight into
      // the B units of layer 0 (identity slot order).  The frame's KE rows are staged in LDS once per tile when the tile lies in one frame.
      static_assert(KE <= 96, "affine table");
      const GLOBAL_AS float4* tabg = (const GLOBAL_AS float4*)a.aff;
      float4* ltab = afftab + wid * 96;
      const bool uni = (a.spf % TILE) == 0 && a.frame_idx == nullptr;
      if (uni) {
        const int m = __builtin_amdgcn_readfirstlane(frame[0]);
        for (int e = lane; e < KE; e += 64) {
          const f32x4_t r = *(const GLOBAL_AS f32x4_t*)(tabg + (size_t)m * KE + e);
          ltab[e] = make_float4(r.x, r.y, r.z, r.w);
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int s = sidx[t] < S_eff ? sidx[t] : S_eff - 1;
        const float x0 = a.x[(size_t)s * 3], x1 = a.x[(size_t)s * 3 + 1], x2_ = a.x[(size_t)s * 3 + 2];
        auto slot_val = [&](int slot) {
          float4 r;
          if (uni) r = ltab[slot];
          else {
            const f32x4_t g4 = *(const GLOBAL_AS f32x4_t*)(tabg + (size_t)frame[t] * KE + slot);
            r = make_float4(g4.x, g4.y, g4.z, g4.w);
          }
          return relu1(r.x * x0 + r.y * x1 + r.z * x2_ + r.w);
        };
#pragma unroll
        for (int g = 0; g < UE; ++g) {
          if constexpr (P::BF16) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = slot_val(16 * g + 8 * h + j);
            emb[t][g] = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
          } else {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = slot_val(2 * (4 * g + e) + h);
            emb[t][g] = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if constexpr (Net::EMB == 2 && !TAN) break;  // formed above
      const int s = sidx[t] < S_eff ? sidx[t] : S_eff - 1;
      if constexpr (Net::EMB == 0 && !TAN) {
        float x[6] = {a.x[(size_t)s * 3], a.x[(size_t)s * 3 + 1], a.x[(size_t)s * 3 + 2], 0.f, 0.f, 0.f};
        if constexpr (Net::AUX3) { x[3] = a.x2[(size_t)s * 3]; x[4] = a.x2[(size_t)s * 3 + 1]; x[5] = a.x2[(size_t)s * 3 + 2]; }
        if constexpr (P::BF16) {
          // bf16 path: sin/cos(2^f x) by angle doubling from one accurate sincos per axis
          // (sin 2a = 2 s c, cos 2a = 1 - 2 s^2).  The recurrence error doubles per octave (<= 2^11 * 6e-8 = 1.2e-4),
          // far below the 4e-3 rounding of the bf16 operand it feeds; the fp32 path below keeps exact sincosf.
          float sv[Net::NFREQ > 0 ? Net::NFREQ : 1][3], cv[Net::NFREQ > 0 ? Net::NFREQ : 1][3];
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) {
            float sn, cs;
            sincosf(x[ax], &sn, &cs);
#pragma unroll
            for (int f = 0; f < Net::NFREQ; ++f) {
              sv[f][ax] = sn; cv[f][ax] = cs;
              sincos_double(sn, cs);
            }
          }
#pragma unroll
          for (int g = 0; g < UE; ++g) {
            unsigned int w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // slots 16g + 8h + 2i, +1  = pair 8g + 4h + i  (h selects one of two compile-time pairs)
              float v0[2], v1[2];
#pragma unroll
              for (int hh = 0; hh < 2; ++hh) {
                constexpr int L = Net::NFREQ;
                const int pair = 8 * g + 4 * hh + i;  // compile-time after unrolling
                if (pair < 3 * L) {
                  const int f = pair / 3, ax = pair - 3 * f;
                  const float wf = a.freq_w ? a.freq_w[f] : 1.0f;
                  v0[hh] = sv[f < L ? f : 0][ax] * wf; v1[hh] = cv[f < L ? f : 0][ax] * wf;
                } else {
                  const int sl = 2 * pair - 6 * L;
                  constexpr int NR = Net::AUX3 ? 6 : 3;  // raw block: [x | aux]
                  v0[hh] = sl < NR ? x[sl < NR ? sl : 0] : 0.f;
                  v1[hh] = sl + 1 < NR ? x[sl + 1 < NR ? sl + 1 : 0] : 0.f;
                }
              }
              w[i] = pack2bf(h ? v0[1] : v0[0], h ? v1[1] : v1[0]);
            }
            emb[t][g] = make_uint4(w[0], w[1], w[2], w[3]);
          }
        } else {
#pragma unroll
          for (int g = 0; g < UE; ++g) {
            float w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // k-step 4g+e, slot 2(4g+e) + h
              float v0, v1;
              emb_pair<Net>(4 * g + e, x, a.freq_w, v0, v1);
              w[e] = h ? v1 : v0;
            }
            emb[t][g] = make_uint4(__float_as_uint(w[0]), __float_as_uint(w[1]), __float_as_uint(w[2]), __float_as_uint(w[3]));
          }
        }
      } else {  // raw channels (tangent mode: KE slots), from the staged tile: row of this lane's sample
        const float* xr = stagef + (NT * n + t) * CINR;
#pragma unroll
        for (int g = 0; g < UE; ++g) {
          if constexpr (P::BF16) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int c = 16 * g + 8 * h + j;
              v[j] = c < CINR ? xr[c] : 0.f;
            }
            emb[t][g] = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
          } else {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int c = 2 * (4 * g + e) + h;
              v[e] = c < CINR ? xr[c] : 0.f;
            }
            emb[t][g] = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
          }
        }
      }
    }
    // store the embedding [slot][sample] for the backward / wgrad (quad-transposed 16-byte stores, see store_tile)
    if constexpr (ST) {
      const int q = n & 3, kq = n >> 2;
      GLOBAL_AS char* base = (GLOBAL_AS char*)a.emb + tile_base_offset<P>(KE, s0, 0);
#pragma unroll
      for (int g = 0; g < UE; ++g) {
        if constexpr (P::BF16) {
          // unit g: slots 16g + 8h + j (j = 0..7); emb[t][g] word (j>>1) half (j&1)
          const unsigned int w0[4] = {emb[0][g].x, emb[0][g].y, emb[0][g].z, emb[0][g].w};
          const unsigned int w1[4] = {emb[1][g].x, emb[1][g].y, emb[1][g].z, emb[1][g].w};
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
        } else {
          // unit g: k-steps 4g+e, slot 2(4g+e) + h
          unsigned int d[4] = {emb[0][g].x, emb[0][g].y, emb[0][g].z, emb[0][g].w};
          quad_transpose(d, q);
          gst16(base + tile_lane_offset<P>(2 * (4 * g + q) + h, kq), d[0], d[1], d[2], d[3]);
        }
      }
    }

    // ---- layers ----
#pragma nounroll
    for (int l = 0; l < Net::NL; ++l)
    sfor<0, Net::NL>([&](auto ri) {
      constexpr int R = decltype(ri)::value;  // representative layer of a kind (see fwd_rep)
      if constexpr (fwd_rep<Net>(R) != R) return;
      constexpr unsigned MEMBERS = fwd_members<Net>(R);  // forced compile-time (otherwise evaluated by the wave!)
      if (!((MEMBERS >> l) & 1u)) return;
      constexpr LS ls = Net::L[R];
      constexpr bool LAST = (R == Net::NL - 1);
      constexpr int MT = pad32(ls.mout) / 32;
      constexpr int GE = ls.ke / P::FPG, GA = ls.kin / P::FPG, G = GE + GA;
      constexpr int GL = G < ACG ? G : ACG, NQ = (GL + 3) / 4;  // A groups shared through LDS / fetched per wave
      const GLOBAL_AS void* Wl = KARG_PTR(FwdK, const void*, W, l);
      const GLOBAL_AS float* bl = KARG_PTR(FwdK, const float*, bias, l);
      const GLOBAL_AS float* pfl = KARG_PTR(FwdK, const float*, pf_bias, l);
      GLOBAL_AS unsigned int* maskl = KARG_PTR(FwdK, unsigned int*, mask, l);
      GLOBAL_AS void* actl = KARG_PTR(FwdK, void*, act, l);
      // inputs from the previous layer: slab -> registers
      uint4 bin[NT][GA > 0 ? GA : 1];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < GA; ++u) bin[t][u] = slab[(t * UW + u) * 64];
      // Software-pipelined M-tile loop: the A groups of tile mt+1 are requested right after the MFMAs of tile mt
      // have issued and BEFORE its activation stores.  vmcnt retires in order and counts stores too, so with the
      // loads ahead of the stores in the queue the next tile never waits for a store acknowledgement.
      // bias (+ per-frame bias) of one M-tile in accumulator layout: feature 32mt + 8i + 4h + (0..3).  Like the A groups it
      // is requested one tile ahead (right after the MFMAs issue, BEFORE the epilogue's stores): a bias load at the top
      // of the tile would sit behind those stores in the in-order vmcnt queue and cost a full store round trip per tile.
      constexpr int NB = (ls.pf != 0 && !TAN) ? NT : 1;
      auto load_bias = [&](int mt, f32x16_t (&bv)[NB]) {
        if constexpr (TAN) {
#pragma unroll
          for (int r = 0; r < 16; ++r) bv[0][r] = 0.f;
        } else {
#pragma unroll
          for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              // pf layers: the per-frame table already contains the shared bias (host contract, lab4d_mlp.h)
              const GLOBAL_AS float* src = (ls.pf != 0) ? pfl + (size_t)frame[t] * (32 * MT) : bl;
              const f32x4_t v = *(const GLOBAL_AS f32x4_t*)(src + 32 * mt + 8 * i + 4 * h);
              bv[t][4 * i + 0] = v.x; bv[t][4 * i + 1] = v.y; bv[t][4 * i + 2] = v.z; bv[t][4 * i + 3] = v.w;
            }
        }
      };
      // ---- MFMA phase of one M-tile: acc = bias + W[mt] x.  `pre` = tile whose A groups / bias are requested behind it
      // (each group's registers are re-loaded as soon as its MFMAs have issued), or -1.
      // rbuf >= 0 (progressive reload): the LDS-shared groups of tile `pre` are read back from buffer rbuf group by group, each
      // right after the MFMAs that consumed the register it overwrites -- the LDS round trip of the next tile's weights hides
      // behind this tile's matrix work instead of sitting between the barrier and the first MFMA of every step.
      auto mfma_tile = [&](auto has_pre, int pre, int rbuf, uint4 (&A)[G], f32x16_t (&bv)[NB], f32x16_t (&acc)[NT]) {
        constexpr bool PRE = decltype(has_pre)::value;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = bv[NB == 1 ? 0 : t];
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
          for (int t = 0; t < NT; ++t) mma_unit<P>(acc[t], A[g], g < GE ? emb[t][g < GE ? g : 0] : bin[t][g >= GE ? g - GE : 0]);
          if constexpr (PRE) {
            if (g >= GL) A[g] = load_a(Wl, G, pre, g, lane);  // groups beyond the LDS-shared ones (g is unrolled)
            else if (rbuf >= 0) A[g] = abuf[(rbuf * ACG + g) * 64 + lane];
          }
        }
        if constexpr (PRE) load_bias(pre, bv);
      };
      // workgroup-shared A groups (see wg_step_barrier): wave w moves groups w, w+4, ... (clamped: a duplicate fetch of the
      // last group keeps the code branch-free when GL is not a multiple of 4)
      auto a_fetch = [&](int mt, uint4 (&stg)[NQ]) {
#ifndef LAB4D_ABL_NOAFETCH
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int g = wid + 4 * i < GL ? wid + 4 * i : GL - 1;
          stg[i] = load_a(Wl, G, mt, g, lane);
        }
#endif
      };
      auto a_stash = [&](int buf, const uint4 (&stg)[NQ]) {
#ifndef LAB4D_ABL_NOAFETCH
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int g = wid + 4 * i < GL ? wid + 4 * i : GL - 1;
          abuf[(buf * ACG + g) * 64 + lane] = stg[i];
        }
#endif
      };
      auto a_grab = [&](int buf, uint4 (&A)[G]) {
#pragma unroll
        for (int g = 0; g < GL; ++g) A[g] = abuf[(buf * ACG + g) * 64 + lane];
      };
      // ---- HBM inputs of an epilogue (tangent mode: the primal's sign bits; colour net: the basefield feature tile) are
      // requested one pipeline step ahead, like the A groups
      unsigned int tan_bits = 0;
      uint4 ext_raw[4];
      auto prefetch = [&](int mt) {
        if constexpr (TAN && ls.relu != 0) tan_bits = maskl[((size_t)tile * MT + mt) * 64 + lane];
        if constexpr (ls.add_ext != 0 && !TAN) load_tile_raw<P>((const GLOBAL_AS void*)a.ext, 32 * MT, s0, mt, lane, ext_raw);
      };
      // ---- epilogue of one M-tile: ReLU (+ sign mask), ext add, activation store, hand-over to the next layer
      // Packed tiles keep their global stores for `flush`, which the step issues AFTER the weight loads of the tile after next:
      // the CU's vector-memory queue is a FIFO, and a weight load queued behind a tile's 5 KB of activation stores (17 KB per CU
      // and step, drained at the HBM rate) came back later than the step that needed it (measured: removing the weight stream
      // made the forward 19 % faster although it is 3 % of the bytes).  Loads first, stores last: a load now only has the
      // stores of the PREVIOUS step in front of it, issued a whole step earlier.
      constexpr bool PACKED = P::BF16 && !TAN && ls.add_ext == 0 && !LAST;
      auto epilogue = [&](int mt, f32x16_t (&acc)[NT], unsigned int (&w)[2][8], unsigned int& wbits) {
        if constexpr (PACKED) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc_fence_if<(want_occ<Net, P>() == 2)>(acc[t]);  // the packing below is inline asm
          // packed-bf16 epilogue (see pk_* helpers): everything after the one fp32 -> bf16 conversion works on the 16 packed dwords
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int k = 0; k < 8; ++k) w[t][k] = pack2bf_op(acc[t][2 * k], acc[t][2 * k + 1]);
          if constexpr (ls.relu != 0) {
#ifndef LAB4D_ABL_NOMASK
            if constexpr (ST) wbits = pk_alive_bits(w);
#endif
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int k = 0; k < 8; ++k) w[t][k] = pk_relu_bf16(w[t][k]);
          }
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 2; ++q) slab[(t * UW + 2 * mt + q) * 64] = make_uint4(w[t][4 * q], w[t][4 * q + 1], w[t][4 * q + 2], w[t][4 * q + 3]);
#if defined(LAB4D_TRSTORE) && !defined(LAB4D_ABL_NOSTORE)
          if constexpr (ST || fwd_any_export<Net>(R)) tr_issue(tr_base + (unsigned)mt * 2048u, trt);  // read back transposed, stored in `flush`
#endif
          return;
        }
        if constexpr (ls.relu != 0) {
          // ReLU + its sign bits (1 dword per lane per tile): the backward masks with these instead of re-reading
          // the whole activation tile (16x less traffic, 31 fewer live registers)
          if constexpr (TAN) {
            const unsigned int bits = tan_bits;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int r = 0; r < 16; ++r)
                acc[t][r] = __uint_as_float(__float_as_uint(acc[t][r]) & (unsigned int)__builtin_amdgcn_sbfe((int)bits, mask_bit<P>(t, r), 1));
          } else {
#ifndef LAB4D_ABL_NOMASK
            if constexpr (ST && !LAST) {
              unsigned int bits = 0;
#pragma unroll
              for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) bits |= (acc[t][r] > 0.f ? 1u : 0u) << mask_bit<P>(t, r);
              maskl[((size_t)tile * MT + mt) * 64 + lane] = bits;
            }
#endif
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[t][r] = relu1(acc[t][r]);
          }
        }
        if constexpr (ls.add_ext != 0 && !TAN) {
          f32x16_t e[NT];
          tile_from_raw<P>(ext_raw, lane, e);
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] += e[t][r];
        }
        if constexpr (ST) {
#ifndef LAB4D_ABL_NOSTORE
          if constexpr (!LAST && ACTS) store_tile<P>(actl, 32 * MT, s0, mt, lane, acc);
#endif
        } else if constexpr (fwd_any_export<Net>(R)) {
          if (actl) store_tile<P>(actl, 32 * MT, s0, mt, lane, acc);  // inference: only the layer another net consumes
        }
        if constexpr (!LAST) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            uint4 u[P::UPT];
            tile_to_units<P>(acc[t], u);
#pragma unroll
            for (int q = 0; q < P::UPT; ++q) slab[(t * UW + P::UPT * mt + q) * 64] = u[q];
          }
        } else if constexpr (head_staged<Net>() && !TAN) {
          // head with many channels (delta-skin logits, features): written lane by lane the (S, COUT) rows cost COUT 4-byte stores per sample at a
          // COUT*4-byte stride; the tile goes through the wave's slab (idle: this layer's inputs are in registers, no layer follows) and leaves
          // as one contiguous, coalesced region
          static_assert(pad32(Net::COUT) == 32, "one row tile");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int f = drow(r, h);
              if (f < Net::COUT) stagef[(NT * n + t) * Net::COUT + f] = acc[t][r];
            }
          __builtin_amdgcn_wave_barrier();
          if (a.out) stage_out(stagef, a.out, (long)s0 * Net::COUT, TILE * Net::COUT, (long)S_eff * Net::COUT - 1, lane);
          __builtin_amdgcn_wave_barrier();
        } else {
          // head: raw outputs (S, COUT) fp32
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int f = 32 * mt + drow(r, h);
              if (f < Net::COUT && sidx[t] < S_eff && a.out) a.out[(size_t)sidx[t] * Net::COUT + f] = acc[t][r];
            }
        }
      };
      auto flush = [&](int mt, const unsigned int (&w)[2][8], unsigned int wbits) {
        if constexpr (PACKED) {
          if constexpr (ST) {
#ifndef LAB4D_ABL_NOMASK
            if constexpr (ls.relu != 0) maskl[((size_t)tile * MT + mt) * 64 + lane] = wbits;
#endif
#ifndef LAB4D_ABL_NOSTORE
#ifndef LAB4D_TRSTORE
            if constexpr (ACTS) store_tile_packed(actl, 32 * MT, s0, mt, lane, w);
#else
            tr_wait(trt);
            if constexpr (ACTS) tr_store(actl, 32 * MT, s0, mt, lane, trt);
#endif
#endif
          } else if constexpr (fwd_any_export<Net>(R)) {
#ifndef LAB4D_TRSTORE
            if (actl) store_tile_packed(actl, 32 * MT, s0, mt, lane, w);  // inference: only the layer another net consumes
#else
            tr_wait(trt);
            if (actl) tr_store(actl, 32 * MT, s0, mt, lane, trt);
#endif
          }
        }
      };
      // Software pipeline over the M-tiles (one wave per SIMD: nothing else hides the epilogue): step k issues the
      // MFMAs of tile k+1 into the second accumulator set and, in the same basic block, runs the epilogue of tile k, so
      // the VALU / LDS / store work of one tile overlaps the matrix pipe of the next.  Accumulator sets alternate, hence
      // the pair loop; MT is a compile-time constant so the tail is resolved statically.
      uint4 A[G], stg[NQ];
      f32x16_t bv[NB];
      f32x16_t acc0[NT], acc1[NT];
      unsigned int pw[2][8], pbits = 0;  // packed tile + ReLU bits waiting for `flush`
      // prologue: tiles 0 and 1 into the two LDS buffers (the barrier in front keeps a fast wave from overwriting groups
      // a slow one still has to read for the previous layer), tile 2 requested into the staging registers
      wg_step_barrier();
      a_fetch(0, stg);
      a_stash(0, stg);
      if constexpr (MT > 1) {
        a_fetch(1, stg);
        a_stash(1, stg);
      }
      a_fetch(MT > 2 ? 2 : MT - 1, stg);
#pragma unroll
      for (int g = GL; g < G; ++g) A[g] = load_a(Wl, G, 0, g, lane);
      load_bias(0, bv);
      prefetch(0);
      wg_step_barrier();
      a_grab(0, A);
      constexpr int NSTEP = MT - 1, NPAIR = NSTEP / 2;
#ifdef LAB4D_PROG_FWD  // measured: -3..-6 % on the backward chains, but the forward chains (which also hold the embedding and the bias tile) spill 26-73 registers with it and get 7-10 % slower: backward only
      // Progressive weight reload.  Step of tile T: the A registers hold tile T; each group is re-read from buffer (T+1)&1
      // (tile T+1, stashed during the previous step) right behind its MFMAs; the staging registers (tile T+2) go to buffer T&1,
      // which was last read during the previous step; tile T+3 is requested.  One barrier per step as before: it orders
      // "everybody has read buffer T&1" before this step's stash and "stash of tile T+1 visible" before this step's reads.
      mfma_tile(std::bool_constant<(MT > 1)>{}, 1, 1, A, bv, acc0);  // tile 0, reloading tile 1 from buffer 1
      if constexpr (MT > 1) {
        wg_step_barrier();  // every wave has grabbed tile 0 out of buffer 0
        a_stash(0, stg);    // tile 2
        a_fetch(MT > 3 ? 3 : MT - 1, stg);
      }
      if constexpr (NPAIR > 0) {
#pragma nounroll
        for (int k = 0; k < 2 * NPAIR; k += 2) {
          wg_step_barrier();
          mfma_tile(std::true_type{}, k + 2 < MT ? k + 2 : MT - 1, 0, A, bv, acc1);  // tile k+1, reloading tile k+2 from buffer 0
          epilogue(k, acc0, pw, pbits);
          prefetch(k + 1);
          a_stash(1, stg);  // tile k+3
          a_fetch(k + 4 < MT ? k + 4 : MT - 1, stg);
          flush(k, pw, pbits);
          wg_step_barrier();
          mfma_tile(std::true_type{}, k + 3 < MT ? k + 3 : MT - 1, 1, A, bv, acc0);  // tile k+2, reloading tile k+3 from buffer 1
          epilogue(k + 1, acc1, pw, pbits);
          prefetch(k + 2 < MT ? k + 2 : MT - 1);
          a_stash(0, stg);  // tile k+4
          a_fetch(k + 5 < MT ? k + 5 : MT - 1, stg);
          flush(k + 1, pw, pbits);
        }
      }
      if constexpr (NSTEP % 2 == 1) {
        wg_step_barrier();
        mfma_tile(std::false_type{}, 0, -1, A, bv, acc1);  // tile MT-1
        epilogue(MT - 2, acc0, pw, pbits);
        flush(MT - 2, pw, pbits);
        prefetch(MT - 1);
        epilogue(MT - 1, acc1, pw, pbits);
        flush(MT - 1, pw, pbits);
      } else {
        epilogue(MT - 1, acc0, pw, pbits);
        flush(MT - 1, pw, pbits);
      }
#else
      mfma_tile(std::bool_constant<(MT > 1)>{}, 1, -1, A, bv, acc0);
      if constexpr (NPAIR > 0) {
#pragma nounroll
        for (int k = 0; k < 2 * NPAIR; k += 2) {
          // step k: A(k+1) is in buffer 1; the staging registers hold A(k+2) (requested during the previous step), which goes
          // to buffer 0 (read last in step k-1); then A(k+3) is requested, and only then tile k's stores are issued
          wg_step_barrier();
          a_grab(1, A);
          mfma_tile(std::true_type{}, k + 2 < MT ? k + 2 : MT - 1, -1, A, bv, acc1);  // tile k+1
          epilogue(k, acc0, pw, pbits);
          prefetch(k + 1);
          a_stash(0, stg);
          a_fetch(k + 3 < MT ? k + 3 : MT - 1, stg);
          flush(k, pw, pbits);
          if constexpr (PACKED) sched_interleave<NT * G, LAB4D_SCHED_NV_FWD, LAB4D_SCHED_FWD_ON>();
          // step k+1
          wg_step_barrier();
          a_grab(0, A);
          mfma_tile(std::true_type{}, k + 3 < MT ? k + 3 : MT - 1, -1, A, bv, acc0);  // tile k+2
          epilogue(k + 1, acc1, pw, pbits);
          prefetch(k + 2 < MT ? k + 2 : MT - 1);
          a_stash(1, stg);
          a_fetch(k + 4 < MT ? k + 4 : MT - 1, stg);
          flush(k + 1, pw, pbits);
          if constexpr (PACKED) sched_interleave<NT * G, LAB4D_SCHED_NV_FWD, LAB4D_SCHED_FWD_ON>();
        }
      }
      if constexpr (NSTEP % 2 == 1) {
        wg_step_barrier();
        a_grab(1, A);
        mfma_tile(std::false_type{}, 0, -1, A, bv, acc1);  // tile MT-1
        epilogue(MT - 2, acc0, pw, pbits);
        flush(MT - 2, pw, pbits);
        if constexpr (PACKED) sched_interleave<NT * G, LAB4D_SCHED_NV_FWD, LAB4D_SCHED_FWD_ON>();
        prefetch(MT - 1);
        epilogue(MT - 1, acc1, pw, pbits);
        flush(MT - 1, pw, pbits);
      } else {
        epilogue(MT - 1, acc0, pw, pbits);
        flush(MT - 1, pw, pbits);
      }
#endif
    });
  }
}

// =================================================================================================
// backward (dgrad) chain
// =================================================================================================
template <class Net>
constexpr int emb_layer_count() {
  int c = 0;
  for (int l = 0; l < Net::NL; ++l) c += Net::L[l].ke ? 1 : 0;
  return c;
}

// AU (EMB == 2 nets only): every tile lies in one frame (spf % TILE == 0, the training shapes) -- the frame's table rows are staged in LDS and the
// table gradient is reduced in registers; false: rows read per sample, element-wise atomics (tiny shapes whose tiles straddle frames)
// DZ = false: no dZ is stored (nobody takes a weight gradient: the eval path's normals differentiate wrt the points only)
template <class Net, class P, bool AU = true, bool DZ = true>
__global__ void __launch_bounds__(256, (want_occ<Net, P, true>())) k_mlp_bwd(BwdK a) {
  static_assert(Net::EMB == 0 || emb_layer_count<Net>() == 1, "raw-input nets may use the input in one layer only");
  constexpr int NT = P::NT, TILE = P::TILE, NL = Net::NL, UW = Slab<Net, P>::UW;
  constexpr int ACG = acache_g<Net, P>();
  __shared__ uint4 slab_all[4 * Slab<Net, P>::UNITS_PER_WAVE];
  // workgroup-shared A groups (see wg_step_barrier): two buffers, two LDS variables (the LDS-DMA build relies on the compiler telling them apart)
  __shared__ uint4 abuf0[ACG * 64];
  __shared__ uint4 abuf1[ACG * 64];
#define LAB4D_ABUF(buf) ((buf) ? abuf1 : abuf0)
  __shared__ float4 afftab[Net::EMB == 2 ? 4 * 96 : 1];  // EMB == 2: the current frame's rows of the affine first layer, one copy per wave
  // EMB == 2: per-lane partial of the table gradient, g_aff[frame][32 mt + drow(r, h)][j] += dz0 * [x; 1]_j over this lane's samples; reduced over
  // the 32 lanes of a half and added to g_aff when the wave's tiles move on to another frame (tiles come in increasing order) and at the end
  constexpr int NGA = (Net::EMB == 2 && AU) ? (Net::KE / 32) * 16 * 4 : 1;
  float gacc[NGA];
#pragma unroll
  for (int i = 0; i < NGA; ++i) gacc[i] = 0.f;
  int gframe = -1;
  auto gram_flush = [&]() {
    if constexpr (Net::EMB == 2 && AU) {
      if (gframe >= 0 && a.g_aff != nullptr) {
#pragma unroll
        for (int i = 0; i < NGA; ++i) {
          float v = gacc[i];
#pragma unroll
          for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off, 64);
          const int mt = i / 64, r = (i >> 2) & 15, j = i & 3;
          if ((threadIdx.x & 31) == 0) atomicAdd(a.g_aff + ((size_t)gframe * Net::KE + 32 * mt + drow(r, (int)((threadIdx.x & 63) >> 5))) * 4 + j, v);
          gacc[i] = 0.f;
        }
      }
    }
  };
  const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint4* slab = slab_all + wid * Slab<Net, P>::UNITS_PER_WAVE + lane;
  float* stagef = reinterpret_cast<float*>(slab_all + wid * Slab<Net, P>::UNITS_PER_WAVE);  // wave-private staging (raw-input nets)
  const int wave = blockIdx.x * 4 + wid, nwaves = gridDim.x * 4;
  const unsigned tr_base = tr_lane_base((unsigned)(size_t)(__attribute__((address_space(3))) uint4*)(slab_all + wid * Slab<Net, P>::UNITS_PER_WAVE), UW, lane);
  TrTile trt;

  for (int tile = wave; tile < a.ntiles; tile += nwaves) {
    const int s0 = tile * TILE;
    int sidx[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) sidx[t] = s0 + NT * n + t;
    float dx[NT][3];  // posenc nets: gradient wrt the 3-vector (partial over this lane's slots)
    float dx2[NT][3];  // ... and wrt the aux 3-vector (nets with AUX3)
#pragma unroll
    for (int t = 0; t < NT; ++t) { dx[t][0] = dx[t][1] = dx[t][2] = 0.f; dx2[t][0] = dx2[t][1] = dx2[t][2] = 0.f; }
    float xs[NT][3];  // EMB == 2: the tile's points
    int frm[NT];      // ... and their frames
    float4* ltab = afftab + wid * 96;
    constexpr bool uni = AU;
    if constexpr (Net::EMB == 2) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int sc = sidx[t] < a.S ? sidx[t] : a.S - 1;
        frm[t] = sc / a.spf;
#pragma unroll
        for (int k = 0; k < 3; ++k) xs[t][k] = ((const GLOBAL_AS float*)a.x)[(size_t)sc * 3 + k];
      }
      if constexpr (uni) {
        const int m = __builtin_amdgcn_readfirstlane(frm[0]);
        if (m != gframe) {
          gram_flush();
          gframe = m;
        }
        for (int e = lane; e < Net::KE; e += 64) {
          const f32x4_t r4 = *(const GLOBAL_AS f32x4_t*)((const GLOBAL_AS float4*)a.aff + (size_t)m * Net::KE + e);
          ltab[e] = make_float4(r4.x, r4.y, r4.z, r4.w);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }

    // ---- head gradient: (S, COUT) fp32 -> accumulator layout -> stored + B units in the slab ----
    {
      f32x16_t g[NT];
      if constexpr (head_staged<Net>()) {  // the tile's (TILE, COUT) rows: one coalesced copy into the (still idle) slab, then row reads (see the forward head)
        __builtin_amdgcn_wave_barrier();
        stage_in<TILE * Net::COUT>(stagef, a.d_out, (long)s0 * Net::COUT, (long)a.S * Net::COUT - 1, lane);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int f = drow(r, h);
            g[t][r] = (f < Net::COUT && sidx[t] < a.S) ? stagef[(NT * n + t) * Net::COUT + (f < Net::COUT ? f : 0)] : 0.f;
          }
        __builtin_amdgcn_wave_barrier();
      } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = drow(r, h);
          g[t][r] = (f < Net::COUT && sidx[t] < a.S) ? a.d_out[(size_t)sidx[t] * Net::COUT + f] : 0.f;
        }
      }
      if constexpr (DZ) store_tile<P>((GLOBAL_AS void*)a.dz[NL - 1], pad32(Net::L[NL - 1].mout), s0, 0, lane, g);  // every dz[l] is required (host-checked): no stores in runtime branches
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        uint4 u[P::UPT];
        tile_to_units<P>(g[t], u);
#pragma unroll
        for (int q = 0; q < P::UPT; ++q) slab[(t * UW + q) * 64] = u[q];
      }
    }

#pragma nounroll
    for (int l = NL - 1; l >= 0; --l)
    sfor<0, NL>([&](auto ri) {
      constexpr int R = NL - 1 - decltype(ri)::value;  // representative layer of a kind (see bwd_rep), NL-1 .. 0
      if constexpr (bwd_rep<Net>(R) != R) return;
      constexpr unsigned MEMBERS = bwd_members<Net>(R);
      if (!((MEMBERS >> l) & 1u)) return;
      constexpr LS ls = Net::L[R];
      constexpr LS lp = Net::L[R > 0 ? R - 1 : 0];
      constexpr int GK = pad32(ls.mout) / P::FPG;        // K units = out features of layer l
      constexpr int GL = GK < ACG ? GK : ACG, NQ = (GL + 3) / 4;  // A groups shared through LDS / fetched per wave
      constexpr int MTE = ls.ke / 32, MTA = ls.kin / 32;  // row tiles: embedding slots, then previous activation
      constexpr bool DO_ACT = (R > 0 && MTA > 0);
      const int lm1 = l > 0 ? l - 1 : 0;
      const GLOBAL_AS void* Wt = KARG_PTR(BwdK, const void*, WT, l);
      const GLOBAL_AS unsigned int* maskp = KARG_PTR(BwdK, const unsigned int*, mask, lm1);
      GLOBAL_AS void* dzp = KARG_PTR(BwdK, void*, dz, lm1);
      uint4 bin[NT][GK];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < GK; ++u) bin[t][u] = slab[(t * UW + u) * 64];

      // MFMA phase of one row tile of W^T: acc = W^T[mt] dz.  pre = row tile whose A groups are requested behind it.
      auto no_hook = [&](auto) {};
      auto mfma_tile = [&](auto has_pre, int pre, int rbuf, uint4 (&A)[GK], f32x16_t (&acc)[NT], auto&& hook) {
        constexpr bool PRE = decltype(has_pre)::value;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        sfor<0, GK>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
#pragma unroll
          for (int t = 0; t < NT; ++t) mma_unit<P>(acc[t], A[g], bin[t][g]);
          if constexpr (PRE) {
            if (g >= GL) A[g] = load_a(Wt, GK, pre, g, lane);  // groups beyond the LDS-shared ones
            else if (rbuf >= 0) A[g] = LAB4D_ABUF(rbuf)[g * 64 + lane];  // progressive reload (see the forward kernel)
          }
          hook(gc);
        });
      };
      auto a_fetch = [&](int mt, uint4 (&stg)[NQ]) {
#ifndef LAB4D_ABL_NOAFETCH
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int g = wid + 4 * i < GL ? wid + 4 * i : GL - 1;
          stg[i] = load_a(Wt, GK, mt, g, lane);
        }
#endif
      };
      auto a_stash = [&](int buf, const uint4 (&stg)[NQ]) {
#ifndef LAB4D_ABL_NOAFETCH
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int g = wid + 4 * i < GL ? wid + 4 * i : GL - 1;
          LAB4D_ABUF(buf)[g * 64 + lane] = stg[i];
        }
#endif
      };
      auto a_grab = [&](int buf, uint4 (&A)[GK]) {
#pragma unroll
        for (int g = 0; g < GL; ++g) A[g] = LAB4D_ABUF(buf)[g * 64 + lane];
      };
      auto a_dma = [&](int mt, int buf) {  // this wave's quarter of row tile mt's shared groups -> LDS buffer `buf` (see a_dma_1k)
        const GLOBAL_AS char* tile = (const GLOBAL_AS char*)Wt + (size_t)(unsigned)(mt * GK) * 1024u;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int g = wid + 4 * i < GL ? wid + 4 * i : GL - 1;
          a_dma_1k(tile + (unsigned)(g * 1024 + lane * 16), (__attribute__((address_space(3))) void*)(LAB4D_ABUF(buf) + g * 64));
        }
      };
      // Software pipeline over N row tiles starting at tile0 (same scheme as the forward chain): step k issues the MFMAs
      // of tile k+1 into the other accumulator set in the same basic block as the epilogue of tile k.
      // pre(j) requests the HBM inputs of epi(j) (mask bits, stored embedding / external gradient tile) one step ahead.
      auto pipeline = [&](auto n_c, int tile0, auto&& pre, auto&& epi, auto&& fl, auto&& prem, auto&& sp, auto&& sp_all, auto nst_c) {
        constexpr int N = decltype(n_c)::value;
        // LDS-DMA weight stream: NST = vector-memory operations a step issues LAST, unconditionally (the 4 dZ tile stores of flush_act; 0 for the
        // embedding pipeline, whose epilogue stores nothing): the per-step wait leaves exactly those in flight
        constexpr bool ADMA = use_adma<P>();
        constexpr int NST = decltype(nst_c)::value;
        if constexpr (N > 0) {
          uint4 A[GK], stg[NQ];
          f32x16_t acc0[NT], acc1[NT];
          unsigned int pw[2][8];  // packed tile waiting for its store (see the forward kernel: weight loads first, stores last)
          wg_step_barrier();  // nobody still reads the buffers for the previous pipeline
          if constexpr (ADMA) {
            a_dma(tile0, 0);
            if constexpr (N > 1) a_dma(tile0 + 1, 1);
          } else {
            a_fetch(tile0, stg);
            a_stash(0, stg);
            if constexpr (N > 1) {
              a_fetch(tile0 + 1, stg);
              a_stash(1, stg);
            }
            a_fetch(tile0 + (N > 2 ? 2 : N - 1), stg);
          }
#pragma unroll
          for (int g = GL; g < GK; ++g) A[g] = load_a(Wt, GK, tile0, g, lane);
          pre(0);
          prem(std::integral_constant<int, 0>{}, 0);  // ReLU sign words: requested TWO steps ahead (see pre_mask)
          prem(std::integral_constant<int, 1>{}, N > 1 ? 1 : 0);
          if constexpr (ADMA) wg_step_barrier_dma<0>();  // once per pipeline: both buffers landed (this also drains the previous layer's last stores)
          else wg_step_barrier();
          a_grab(0, A);
          constexpr int NSTEP = N - 1, NPAIR = NSTEP / 2;
#ifndef LAB4D_ABL_NOPROG
          if constexpr (ADMA && N > 1) {
            wg_step_barrier();                      // every wave has grabbed tile 0 out of buffer 0 ...
            a_dma(tile0 + (N > 2 ? 2 : N - 1), 0);  // ... so tile 2 is requested into it BEFORE tile 0's matrix work: a whole step of lead
          }
          mfma_tile(std::bool_constant<(N > 1)>{}, tile0 + 1, 1, A, acc0, no_hook);  // tile 0, reloading tile 1 from buffer 1
          if constexpr (N > 1 && !ADMA) {
            wg_step_barrier();  // every wave has grabbed tile 0 out of buffer 0
            a_stash(0, stg);    // tile 2
            a_fetch(tile0 + (N > 3 ? 3 : N - 1), stg);
          }
          if constexpr (NPAIR > 0) {
#pragma nounroll
            for (int k = 0; k < 2 * NPAIR; k += 2) {
              if constexpr (ADMA) {
                // step of tile k+1: its reload source (tile k+2 in buffer 0) was requested one step ago by every wave -> landed before anyone passes;
                // buffer 1 (tile k+1) was read out during the previous step -> tile k+3 goes there now
                if (k == 0) wg_step_barrier_dma<0>(); else wg_step_barrier_dma<NST>();
                a_dma(tile0 + (k + 3 < N ? k + 3 : N - 1), 1);
                mfma_tile(std::true_type{}, tile0 + (k + 2 < N ? k + 2 : N - 1), 0, A, acc1, no_hook);
                epi(k, acc0, pw);
                pre(k + 1);
                prem(std::integral_constant<int, 0>{}, k + 2 < N ? k + 2 : N - 1);
                fl(k, pw);
                if constexpr (P::BF16) sched_interleave<NT * GK, LAB4D_SCHED_NV, sched_il_bwd<Net>()>();
                wg_step_barrier_dma<NST>();
                a_dma(tile0 + (k + 4 < N ? k + 4 : N - 1), 0);
                mfma_tile(std::true_type{}, tile0 + (k + 3 < N ? k + 3 : N - 1), 1, A, acc0, no_hook);
                epi(k + 1, acc1, pw);
                pre(k + 2 < N ? k + 2 : N - 1);
                prem(std::integral_constant<int, 1>{}, k + 3 < N ? k + 3 : N - 1);
                fl(k + 1, pw);
                if constexpr (P::BF16) sched_interleave<NT * GK, LAB4D_SCHED_NV, sched_il_bwd<Net>()>();
                continue;
              }
              wg_step_barrier();
#ifdef LAB4D_TRSPREAD  // the epilogue first in program order: its slab writes precede the transposing reads of the hook
              epi(k, acc0, pw);
              mfma_tile(std::true_type{}, tile0 + (k + 2 < N ? k + 2 : N - 1), 0, A, acc1, [&](auto gc) { sp(k, gc); });
#else
              mfma_tile(std::true_type{}, tile0 + (k + 2 < N ? k + 2 : N - 1), 0, A, acc1, no_hook);
              epi(k, acc0, pw);
#endif
              pre(k + 1);
              prem(std::integral_constant<int, 0>{}, k + 2 < N ? k + 2 : N - 1);
              a_stash(1, stg);
              a_fetch(tile0 + (k + 4 < N ? k + 4 : N - 1), stg);
              fl(k, pw);
              if constexpr (P::BF16) sched_interleave<NT * GK, LAB4D_SCHED_NV, sched_il_bwd<Net>()>();
              wg_step_barrier();
#ifdef LAB4D_TRSPREAD
              epi(k + 1, acc1, pw);
              mfma_tile(std::true_type{}, tile0 + (k + 3 < N ? k + 3 : N - 1), 1, A, acc0, [&](auto gc) { sp(k + 1, gc); });
#else
              mfma_tile(std::true_type{}, tile0 + (k + 3 < N ? k + 3 : N - 1), 1, A, acc0, no_hook);
              epi(k + 1, acc1, pw);
#endif
              pre(k + 2 < N ? k + 2 : N - 1);
              prem(std::integral_constant<int, 1>{}, k + 3 < N ? k + 3 : N - 1);
              a_stash(0, stg);
              a_fetch(tile0 + (k + 5 < N ? k + 5 : N - 1), stg);
              fl(k + 1, pw);
              if constexpr (P::BF16) sched_interleave<NT * GK, LAB4D_SCHED_NV, sched_il_bwd<Net>()>();
            }
          }
          if constexpr (NSTEP % 2 == 1) {
            if constexpr (ADMA) {
              if constexpr (NPAIR == 0) wg_step_barrier_dma<0>(); else wg_step_barrier_dma<NST>();  // (the last reload already happened: only the lock-step matters)
            } else wg_step_barrier();
#ifdef LAB4D_TRSPREAD
            epi(N - 2, acc0, pw);
            mfma_tile(std::false_type{}, 0, -1, A, acc1, [&](auto gc) { sp(N - 2, gc); });
#else
            mfma_tile(std::false_type{}, 0, -1, A, acc1, no_hook);
            epi(N - 2, acc0, pw);
#endif
            fl(N - 2, pw);
            if constexpr (P::BF16) sched_interleave<NT * GK, LAB4D_SCHED_NV, sched_il_bwd<Net>()>();
            pre(N - 1);
            epi(N - 1, acc1, pw);
            fl(N - 1, pw);
            sp_all(N - 1);  // no matrix work left in this layer to hide the last tile's pieces behind
          } else {
            epi(N - 1, acc0, pw);
            fl(N - 1, pw);
            sp_all(N - 1);
          }
#else
          mfma_tile(std::bool_constant<(N > 1)>{}, tile0 + 1, -1, A, acc0, no_hook);
          if constexpr (NPAIR > 0) {
#pragma nounroll
            for (int k = 0; k < 2 * NPAIR; k += 2) {
              wg_step_barrier();
              a_grab(1, A);
              mfma_tile(std::true_type{}, tile0 + (k + 2 < N ? k + 2 : N - 1), -1, A, acc1, no_hook);
              epi(k, acc0, pw);
              pre(k + 1);
              a_stash(0, stg);
              a_fetch(tile0 + (k + 3 < N ? k + 3 : N - 1), stg);
              fl(k, pw);
              wg_step_barrier();
              a_grab(0, A);
              mfma_tile(std::true_type{}, tile0 + (k + 3 < N ? k + 3 : N - 1), -1, A, acc0, no_hook);
              epi(k + 1, acc1, pw);
              pre(k + 2 < N ? k + 2 : N - 1);
              a_stash(1, stg);
              a_fetch(tile0 + (k + 4 < N ? k + 4 : N - 1), stg);
              fl(k + 1, pw);
            }
          }
          if constexpr (NSTEP % 2 == 1) {
            wg_step_barrier();
            a_grab(1, A);
            mfma_tile(std::false_type{}, 0, -1, A, acc1, no_hook);
            epi(N - 2, acc0, pw);
            fl(N - 2, pw);
            pre(N - 1);
            epi(N - 1, acc1, pw);
            fl(N - 1, pw);
          } else {
            epi(N - 1, acc0, pw);
            fl(N - 1, pw);
          }
#endif
        }
      };
      uint4 raw[4];            // prefetched tile: stored embedding (epi_emb) or external gradient (epi_act)
      unsigned int mbits = 0;  // prefetched ReLU sign bits
      // The sign words come from HBM (written by the forward pass a whole chunk earlier).  Requested one step ahead (~0.4 us of
      // matrix work) they arrived later than the epilogue that needs them; they are requested TWO steps ahead into a two-deep ring
      // (even / odd tile, one more register): basefield backward -3.5 %, colour -5 %, feature -6 %, skin -7 % (a build with the
      // mask work removed altogether bounds the gain at -9 .. -19 %).  -DLAB4D_ABL_MASK1 restores the one-step form.
      unsigned int mb0 = 0, mb1 = 0;
#if !defined(LAB4D_ABL_MASK1) && !defined(LAB4D_ABL_NOPROG)
#define LAB4D_MASK_RING 1
#endif
#if (defined(LAB4D_MASK_RING) || defined(LAB4D_TRSPREAD)) && defined(LAB4D_ABL_NOPROG)
#error "LAB4D_MASK_RING / LAB4D_TRSPREAD are wired into the progressive-reload pipeline only"
#endif
      auto no_prem = [&](auto, int) {};
      auto pre_mask = [&](auto par_c, int j) {
#ifdef LAB4D_MASK_RING
        if constexpr (lp.relu != 0) {
          const unsigned int v = maskp[((size_t)tile * (pad32(lp.mout) / 32) + j) * 64 + lane];
          if constexpr (decltype(par_c)::value == 0) mb0 = v; else mb1 = v;
        }
#endif
      };
      auto pre_emb = [&](int mt) {
        if constexpr (Net::EMB != 1) load_tile_raw<P>((const GLOBAL_AS void*)a.emb, Net::KE, s0, mt, lane, raw);
      };
      auto pre_act = [&](int j) {
#ifndef LAB4D_MASK_RING
        if constexpr (lp.relu != 0) mbits = maskp[((size_t)tile * (pad32(lp.mout) / 32) + j) * 64 + lane];
#endif
        if constexpr (lp.ext_grad != 0) load_tile_raw<P>((const GLOBAL_AS void*)a.ext_gin, pad32(lp.mout), s0, j, lane, raw);
      };
      // (a) gradient wrt the embedding slots -> input gradient
      auto no_flush = [&](int, const unsigned int (&)[2][8]) {};
      auto epi_emb = [&](int mt, f32x16_t (&acc)[NT], unsigned int (&)[2][8]) {
        if constexpr (Net::EMB == 0) {
          f32x16_t e[NT];
          tile_from_raw<P>(raw, lane, e);
          constexpr int L = Net::NFREQ;
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int slot = 32 * mt + drow(r, h);
              const float gv = acc[t][r];
              if (slot < 6 * L) {
                const int pair = slot >> 1, f = pair / 3, ax = pair - 3 * f;
                // d/dx [w sin(2^f x)] = 2^f (w cos) ; d/dx [w cos(2^f x)] = -2^f (w sin): partner slot = register r^1
                const float partner = e[t][r ^ 1];
                const float c = ldexpf((r & 1) ? -partner : partner, f) * gv;
                dx[t][0] += ax == 0 ? c : 0.f;
                dx[t][1] += ax == 1 ? c : 0.f;
                dx[t][2] += ax == 2 ? c : 0.f;
              } else if (slot < 6 * L + 3) {
                const int ax = slot - 6 * L;
                dx[t][0] += ax == 0 ? gv : 0.f;
                dx[t][1] += ax == 1 ? gv : 0.f;
                dx[t][2] += ax == 2 ? gv : 0.f;
              } else if (Net::AUX3 && slot < 6 * L + 6) {
                const int ax = slot - 6 * L - 3;
                dx2[t][0] += ax == 0 ? gv : 0.f;
                dx2[t][1] += ax == 1 ? gv : 0.f;
                dx2[t][2] += ax == 2 ? gv : 0.f;
              }
            }
        } else if constexpr (Net::EMB == 2) {
          // adjoint of slot = relu(aff[frame][slot] . [x; 1]): the stored slot value (> 0 or 0) is the ReLU mask; the point gradient and the
          // table gradient are both fp32 FMAs on the accumulator tile -- no (S, KE) gradient tensor, no weight-gradient launch for this layer
          f32x16_t e[NT];
          tile_from_raw<P>(raw, lane, e);
          auto body = [&](auto uni_c) {
            constexpr bool U = decltype(uni_c)::value;
            sfor<0, 16>([&](auto rc) {
              constexpr int r = decltype(rc)::value;
              const int slot = 32 * mt + drow(r, h);
              float4 row;
              if constexpr (U) row = ltab[slot];
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                if constexpr (!U) {
                  const f32x4_t g4 = *(const GLOBAL_AS f32x4_t*)((const GLOBAL_AS float4*)a.aff + (size_t)frm[t] * Net::KE + slot);
                  row = make_float4(g4.x, g4.y, g4.z, g4.w);
                }
                const float dz = e[t][r] > 0.f ? acc[t][r] : 0.f;
                dx[t][0] += row.x * dz;
                dx[t][1] += row.y * dz;
                dx[t][2] += row.z * dz;
                if constexpr (U) {
                  // mt is a runtime value of the pipeline loop: the accumulator block is selected by compile-time comparison (registers, not scratch)
                  sfor<0, Net::KE / 32>([&](auto mc) {
                    constexpr int mm = decltype(mc)::value;
                    if (mt == mm) {
                      gacc[(mm * 16 + r) * 4 + 0] += dz * xs[t][0];
                      gacc[(mm * 16 + r) * 4 + 1] += dz * xs[t][1];
                      gacc[(mm * 16 + r) * 4 + 2] += dz * xs[t][2];
                      gacc[(mm * 16 + r) * 4 + 3] += dz;
                    }
                  });
                } else if (a.g_aff != nullptr && dz != 0.f && sidx[t] < a.S) {
                  float* gp = a.g_aff + ((size_t)frm[t] * Net::KE + slot) * 4;
                  atomicAdd(gp + 0, dz * xs[t][0]);
                  atomicAdd(gp + 1, dz * xs[t][1]);
                  atomicAdd(gp + 2, dz * xs[t][2]);
                  atomicAdd(gp + 3, dz);
                }
              }
            });
          };
          body(std::bool_constant<uni>{});
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int c = 32 * mt + drow(r, h);
              if (c < Net::CIN) stagef[(NT * n + t) * Net::CIN + c] = acc[t][r];  // staged: copied out coalesced below
            }
        }
      };
      // (b) gradient wrt the previous layer's output -> masked dZ_{l-1}
      auto epi_act = [&](int j, f32x16_t (&acc)[NT], unsigned int (&w)[2][8]) {
#ifdef LAB4D_MASK_RING
        const unsigned int bits = (j & 1) ? mb1 : mb0;
#else
        const unsigned int bits = mbits;
#endif
        if constexpr (P::BF16) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc_fence_if<(want_occ<Net, P, true>() == 2)>(acc[t]);  // the packed path converts with inline asm
        }
        if constexpr (lp.ext_grad != 0) {
          f32x16_t eg[NT];
          tile_from_raw<P>(raw, lane, eg);
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] += eg[t][r];
        }
        if constexpr (lp.add_ext != 0) {
          store_tile<P>((GLOBAL_AS void*)a.ext_gout, pad32(lp.mout), s0, j, lane, acc);  // y = relu(z) + ext  ->  dL/dext = dL/dy
        }
        if constexpr (P::BF16) {
          // packed path: one conversion, the ReLU mask and the zero of the padded tail samples applied to the packed pairs with
          // v_pk_mul_lo_u16 by 0/1 halves (3 ops per pair instead of 2 per value), packed store and hand-over
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int k = 0; k < 8; ++k) w[t][k] = pack2bf_op(acc[t][2 * k], acc[t][2 * k + 1]);
#ifndef LAB4D_ABL_NOMASK
          {
            unsigned int alive = lp.relu != 0 ? bits : 0xffffffffu;
#pragma unroll
            for (int t = 0; t < 2; ++t)
              if (sidx[t] >= a.S) alive &= ~(0x00ff00ffu << (8 * t));
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int k = 0; k < 8; ++k) w[t][k] = pk_mask_bf16(w[t][k], pk_m01(alive, t, k));
          }
#endif
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 2; ++q) slab[(t * UW + 2 * j + q) * 64] = make_uint4(w[t][4 * q], w[t][4 * q + 1], w[t][4 * q + 2], w[t][4 * q + 3]);
#if defined(LAB4D_TRSTORE) && !defined(LAB4D_TRSPREAD) && !defined(LAB4D_ABL_NOSTORE)
          tr_issue(tr_base + (unsigned)j * 2048u, trt);
#endif
        } else {
          // fp32 tiles: ReLU mask and the zero of the padded tail samples in one AND per value (v_bfe_i32 + v_and)
          unsigned int keep = lp.relu != 0 ? bits : 0xffffffffu;
#pragma unroll
          for (int t = 0; t < NT; ++t)
            if (sidx[t] >= a.S) keep &= ~(0xffffu << (16 * t));
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[t][r] = __uint_as_float(__float_as_uint(acc[t][r]) & (unsigned int)__builtin_amdgcn_sbfe((int)keep, 16 * t + r, 1));
          if constexpr (DZ) store_tile<P>(dzp, pad32(lp.mout), s0, j, lane, acc);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            uint4 u[P::UPT];
            tile_to_units<P>(acc[t], u);
#pragma unroll
            for (int q = 0; q < P::UPT; ++q) slab[(t * UW + P::UPT * j + q) * 64] = u[q];
          }
        }
      };
      auto flush_act = [&](int j, const unsigned int (&w)[2][8]) {
        if constexpr (P::BF16) {
#if !defined(LAB4D_ABL_NOSTORE) && !defined(LAB4D_TRSPREAD)
#ifndef LAB4D_TRSTORE
          if constexpr (DZ) store_tile_packed(dzp, pad32(lp.mout), s0, j, lane, w);
#else
          tr_wait(trt);
          tr_store(dzp, pad32(lp.mout), s0, j, lane, trt);
#endif
#endif
        }
      };
      // LAB4D_TRSPREAD: the dZ tile j (packed, in the slab since epi_act) leaves in four pieces between the MFMAs of the second half
      // of the step: piece qa is read at group I(qa) and stored at St(qa) = I(qa) + D; two piece buffers alternate
      TrPiece tp0, tp1;
      auto sp_none = [&](int, auto) {};
      auto sp_all_none = [&](int) {};
      auto sp_act = [&](int j, auto gc) {
#if defined(LAB4D_TRSPREAD) && !defined(LAB4D_ABL_NOSTORE)
        if constexpr (P::BF16) {
          constexpr int g = decltype(gc)::value;
          constexpr int D = GK >= 16 ? GK / 8 : 1;
          if constexpr (GK >= 8) {
            sfor<0, 4>([&](auto qc) {
              constexpr int qa = decltype(qc)::value;
              if constexpr (g == GK - 1 - D * (3 - qa)) trp_store<qa>(dzp, pad32(lp.mout), s0, j, lane, (qa & 1) ? tp1 : tp0);
            });
            sfor<0, 4>([&](auto qc) {
              constexpr int qa = decltype(qc)::value;
              if constexpr (g == GK - 1 - D * (3 - qa) - D) trp_issue<qa>(tr_base + (unsigned)j * 2048u, (qa & 1) ? tp1 : tp0);
            });
          } else if constexpr (g == GK - 1) {  // narrow layers: too few groups to spread over
            tr_issue(tr_base + (unsigned)j * 2048u, trt);
            tr_wait(trt);
            tr_store(dzp, pad32(lp.mout), s0, j, lane, trt);
          }
        }
#endif
      };
      auto sp_all_act = [&](int j) {
#if defined(LAB4D_TRSPREAD) && !defined(LAB4D_ABL_NOSTORE)
        if constexpr (P::BF16) {
          tr_issue(tr_base + (unsigned)j * 2048u, trt);
          tr_wait(trt);
          tr_store(dzp, pad32(lp.mout), s0, j, lane, trt);
        }
#endif
      };
      // embedding row tiles come first in W^T; they are skipped when no input gradient is wanted
      if constexpr (MTE > 0) {
        if (a.d_x != nullptr) {
          pipeline(std::integral_constant<int, MTE>{}, 0, pre_emb, epi_emb, no_flush, no_prem, sp_none, sp_all_none, std::integral_constant<int, 0>{});
          // raw-input nets: the (TILE, CIN) input-gradient tile sits in the wave's staging area (the slab is idle while the
          // last layer's embedding tiles are processed); one contiguous coalesced copy, rows >= S dropped
          if constexpr (Net::EMB == 1) stage_out(stagef, a.d_x, (long)s0 * Net::CIN, TILE * Net::CIN, (long)a.S * Net::CIN - 1, lane);
        }
      }
      if constexpr (DO_ACT) pipeline(std::integral_constant<int, MTA>{}, MTE, pre_act, epi_act, flush_act, pre_mask, sp_act, sp_all_act,
                                     std::integral_constant<int, (DZ ? bwd_step_stores<P>() : 0)>{});
    });

    if constexpr (Net::EMB != 1) {
      if (a.d_x) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
          for (int k = 0; k < 3; ++k) dx[t][k] += __shfl_xor(dx[t][k], 32, 64);
          if (h == 0 && sidx[t] < a.S) {
            a.d_x[(size_t)sidx[t] * 3 + 0] = dx[t][0];
            a.d_x[(size_t)sidx[t] * 3 + 1] = dx[t][1];
            a.d_x[(size_t)sidx[t] * 3 + 2] = dx[t][2];
          }
          if constexpr (Net::AUX3) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dx2[t][k] += __shfl_xor(dx2[t][k], 32, 64);
            if (h == 0 && sidx[t] < a.S && a.d_x2) {
              a.d_x2[(size_t)sidx[t] * 3 + 0] = dx2[t][0];
              a.d_x2[(size_t)sidx[t] * 3 + 1] = dx2[t][1];
              a.d_x2[(size_t)sidx[t] * 3 + 2] = dx2[t][2];
            }
          }
        }
      }
    }
  }
  gram_flush();
#undef LAB4D_ABUF
}

// launchers implemented by each mlp_inst_<net>.hip
template <class Net>
int launch_mlp_fwd(int precision, const FwdK& k, int S, hipStream_t st);
template <class Net>
int launch_mlp_bwd(int precision, const BwdK& k, int S, hipStream_t st);
template <class Net>
int launch_mlp_fwd_tangent(int precision, const FwdK& k, int S, hipStream_t st);

// persistent grid: as many 4-wave workgroups as fit the chip at once (one per CU for the wide nets, whose slabs take most of the
// CU's LDS; two for the narrow ones), asked of the runtime once per kernel
template <auto kernel>
inline int mlp_grid(int ntiles) {
  static int per_cu = 0, n_cu = 0;  // per kernel instantiation
  if (per_cu == 0) {
    int nb = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, 0) != hipSuccess || nb < 1) nb = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    per_cu = nb > 2 ? 2 : nb;
  }
  int g = (ntiles + 3) / 4;
  if (g > n_cu * per_cu) g = n_cu * per_cu;
  static const int grid_env = getenv("LAB4D_CHAIN_GRID") ? atoi(getenv("LAB4D_CHAIN_GRID")) : 0;  // kernel experiments: a resident grid of this many workgroups (part of the chip)
  if (grid_env > 0 && g > grid_env * per_cu) g = grid_env * per_cu;
  return g < 1 ? 1 : g;
}
#define LAB4D_MLP_LAUNCH(KERNEL, k, st)                                                              \
  do {                                                                                               \
    constexpr auto kfn = &KERNEL;                                                                    \
    hipLaunchKernelGGL(kfn, dim3(mlp_grid<kfn>((k).ntiles)), dim3(256), 0, st, k);                   \
  } while (0)

}  // namespace lab4d
#include "mlp_kernels_h.hpp"
#include "mlp_kernels_ws.hpp"
namespace lab4d {
// the point-gradient-only modes (forward: masks + embedding only; backward: no dZ) are instantiated for the sdf basefields: the eval path's normals
template <class Net>
constexpr bool dx_only_ok() { return Net::ID == LAB4D_NET_FG_BASE || Net::ID == LAB4D_NET_BG_BASE; }
// the weights-stationary family (mlp_kernels_ws.hpp) serves the training-mode, inference-mode and point-gradient-only forward and backward of the
// 256-wide posenc nets; returns false when the launch is not its business (other nets, fp32, LAB4D_WS=0)
template <class Net>
inline bool launch_ws_bwd(const BwdK& k0, hipStream_t st) {
  if constexpr (ws_ok<Net>()) {
    if (!ws_enabled()) return false;
    if (!k0.dz[0]) {  // point gradient only (host-checked: the sdf basefields, d_x given)
      if constexpr (dx_only_ok<Net>()) {
        hipLaunchKernelGGL((k_mlp_bwd_ws<Net, false>), dim3(mlp_grid_ws(k0.S_pad / WS_TILE)), dim3(512), 0, st, k0);
        return true;
      }
      return false;
    }
    hipLaunchKernelGGL((k_mlp_bwd_ws<Net, true>), dim3(mlp_grid_ws(k0.S_pad / WS_TILE)), dim3(512), 0, st, k0);
    return true;
  } else {
    return false;
  }
}
template <class Net>
inline bool launch_ws_fwd(const FwdK& k0, hipStream_t st) {
  if constexpr (ws_ok<Net>()) {
    if (!ws_enabled()) return false;
    FwdK k = k0;
    if (k.emb && !k.act[0]) {  // point-gradient-only mode (host-checked: the sdf basefields)
      if constexpr (dx_only_ok<Net>()) {
        hipLaunchKernelGGL((k_mlp_fwd_ws<Net, true, false>), dim3(mlp_grid_ws(k.S_pad / WS_TILE)), dim3(512), 0, st, k);
        return true;
      }
      return false;
    }
    if (k.emb) hipLaunchKernelGGL((k_mlp_fwd_ws<Net, true>), dim3(mlp_grid_ws(k.S_pad / WS_TILE)), dim3(512), 0, st, k);
    else hipLaunchKernelGGL((k_mlp_fwd_ws<Net, false>), dim3(mlp_grid_ws(k.S_pad / WS_TILE)), dim3(512), 0, st, k);
    return true;
  } else {
    return false;
  }
}
// LAB4D_BWD_H=1 routes the 256-wide posenc nets to the 8-wave / 32-sample backward chain (mlp_kernels_h.hpp): parity-green but
// measured SLOWER than the 4-wave kernel (9.12 vs 7.93 ms per 4.2 M samples), so it is off by default
template <class Net>
constexpr bool use_bwd_h() { return Net::EMB == 0 && net_wmax<Net>() == 256; }
inline bool bwd_h_enabled() {
  static const int on = getenv("LAB4D_BWD_H") ? atoi(getenv("LAB4D_BWD_H")) : 0;
  return on != 0;
}


#define LAB4D_MLP_INSTANTIATE(Net)                                                                                        \
  namespace lab4d {                                                                                                       \
  template <>                                                                                                             \
  int launch_mlp_fwd<Net>(int precision, const FwdK& k0, int S, hipStream_t st) {                                         \
    FwdK k = k0;                                                                                                          \
    if (precision == LAB4D_PREC_BF16) {                                                                                   \
      if (launch_ws_fwd<Net>(k, st)) return check_launch("mlp_forward");                                                  \
      k.ntiles = k.S_pad / PBF16::TILE; /* padded tail tiles are processed too: they zero-fill dz */                                                                                  \
      if (k.emb && !k.act[0]) {                                                                                           \
        if constexpr (dx_only_ok<Net>()) LAB4D_MLP_LAUNCH((k_mlp_fwd<Net, PBF16, false, true, false>), k, st);            \
      } else if (k.emb) LAB4D_MLP_LAUNCH((k_mlp_fwd<Net, PBF16, false, true>), k, st);  \
      else LAB4D_MLP_LAUNCH((k_mlp_fwd<Net, PBF16, false, false>), k, st);       \
    } else if (precision == LAB4D_PREC_F32) {                                                                             \
      k.ntiles = k.S_pad / PF32::TILE;                                                                                   \
      if (k.emb && !k.act[0]) {                                                                                           \
        if constexpr (dx_only_ok<Net>()) LAB4D_MLP_LAUNCH((k_mlp_fwd<Net, PF32, false, true, false>), k, st);             \
      } else if (k.emb) LAB4D_MLP_LAUNCH((k_mlp_fwd<Net, PF32, false, true>), k, st);   \
      else LAB4D_MLP_LAUNCH((k_mlp_fwd<Net, PF32, false, false>), k, st);        \
    } else {                                                                                                              \
      set_error("mlp_forward: bad precision %d", precision);                                                              \
      return LAB4D_EINVAL;                                                                                                \
    }                                                                                                                     \
    return check_launch("mlp_forward");                                                                                   \
  }                                                                                                                       \
  template <>                                                                                                             \
  int launch_mlp_bwd<Net>(int precision, const BwdK& k0, int S, hipStream_t st) {                                         \
    BwdK k = k0;                                                                                                          \
    if (precision == LAB4D_PREC_BF16) {                                                                                   \
      if (launch_ws_bwd<Net>(k, st)) return check_launch("mlp_backward");                                                 \
      k.ntiles = k.S_pad / PBF16::TILE; /* padded tail tiles are processed too: they zero-fill dz */                                                                                  \
      if constexpr (use_bwd_h<Net>()) {                                                                                   \
        if (bwd_h_enabled()) {                                                                                            \
          hipLaunchKernelGGL((k_mlp_bwd_h<Net>), dim3(mlp_grid_h(k.S_pad / 32)), dim3(512), 0, st, k);                    \
          return check_launch("mlp_backward");                                                                            \
        }                                                                                                                 \
      }                                                                                                                   \
      if constexpr (Net::EMB == 2) {                                                                                      \
        if (k.spf % PBF16::TILE) LAB4D_MLP_LAUNCH((k_mlp_bwd<Net, PBF16, false>), k, st);                                 \
        else LAB4D_MLP_LAUNCH((k_mlp_bwd<Net, PBF16, true>), k, st);                                                      \
      } else if (!k.dz[0]) {                                                                                              \
        if constexpr (dx_only_ok<Net>()) LAB4D_MLP_LAUNCH((k_mlp_bwd<Net, PBF16, true, false>), k, st);                   \
      } else LAB4D_MLP_LAUNCH((k_mlp_bwd<Net, PBF16>), k, st);                         \
    } else if (precision == LAB4D_PREC_F32) {                                                                             \
      k.ntiles = k.S_pad / PF32::TILE;                                                                                   \
      if constexpr (Net::EMB == 2) {                                                                                      \
        if (k.spf % PF32::TILE) LAB4D_MLP_LAUNCH((k_mlp_bwd<Net, PF32, false>), k, st);                                   \
        else LAB4D_MLP_LAUNCH((k_mlp_bwd<Net, PF32, true>), k, st);                                                       \
      } else if (!k.dz[0]) {                                                                                              \
        if constexpr (dx_only_ok<Net>()) LAB4D_MLP_LAUNCH((k_mlp_bwd<Net, PF32, true, false>), k, st);                    \
      } else LAB4D_MLP_LAUNCH((k_mlp_bwd<Net, PF32>), k, st);                          \
    } else {                                                                                                              \
      set_error("mlp_backward: bad precision %d", precision);                                                             \
      return LAB4D_EINVAL;                                                                                                \
    }                                                                                                                     \
    return check_launch("mlp_backward");                                                                                  \
  }                                                                                                                       \
  }

}  // namespace lab4d
