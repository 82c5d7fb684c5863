/*
 * lab4d_mlp.h -- fused positional-encoding + MLP-stack entry points of liblab4d_hip.so.
 * Included by lab4d_hip.h.
 *
 * Replaces, for the per-sample networks of the Lab4D field (paths relative to lab4d/):
 *   nnutils/embedding.py:69-125   PosEmbedding.forward (incl. the annealing window)
 *   nnutils/base.py:65-78         BaseMLP.forward  (Linear+ReLU stack, skip = cat([x, out]))
 *   nnutils/base.py:123-150       CondMLP.forward  (per-frame code appended to the input)
 * as used by NeRF.forward (nerf.py:167-215), VisField.forward (visibility.py:53-63),
 * FeatureNeRF.compute_feat (feature.py:136-150), SkinningField.forward (skinning.py:108-119) and
 * DenseWarp.forward (warping.py:143-170).
 *
 * Execution model (csrc/mlp.hip): one wavefront owns a tile of 64 (bf16) / 32 (fp32) samples and
 * carries their activations through ALL layers in registers: the 32x32 MFMA accumulator layout of
 * layer l is, by construction of the packed weights, exactly the B-operand layout of layer l+1,
 * so no LDS, no barrier and no cross-lane traffic is needed between layers.  Weights stream from
 * L2 as pre-packed 1-KiB A-fragments (coalesced 16 B per lane).  Per-frame conditioning inputs
 * (instance code, appearance code, time embedding: constant over the samples of a frame) are folded
 * by the host into a per-frame bias, so the kernel never materialises the (S, C) broadcast the
 * reference builds (base.py:139-146).
 *
 * Training stores every post-activation in a [feature][sample] layout (bf16 or fp32); the
 * backward chain kernel re-reads it for the ReLU masks and writes dZ in the same layout; the
 * weight-gradient kernel contracts the two over samples with MFMA (both operands K-contiguous).
 */
#ifndef LAB4D_MLP_H
#define LAB4D_MLP_H

#include <stdint.h>

#define LAB4D_MLP_MAX_LAYERS 12

/* networks (compile-time layer tables live in csrc/mlp_nets.hpp; lab4d_mlp_describe returns them) */
#define LAB4D_NET_FG_BASE 0   /* posenc10 -> basefield (8+1 layers, skip at 4) -> sdf head          */
#define LAB4D_NET_FG_COLOR 1  /* posenc12 -> colorfield (2+1) (+ basefield feature) -> rgb (2)      */
#define LAB4D_NET_VIS 2       /* posenc10 -> 64 -> 64 -> 1                                           */
#define LAB4D_NET_FEAT 3      /* posenc6  -> feature_field (5 layers W=128, skip at 4) -> 16         */
#define LAB4D_NET_SKIN 4      /* raw 75 bone coords -> 64 -> 64 -> 25 (delta skinning weights)       */
#define LAB4D_NET_DENSE 5     /* posenc6 -> 256 -> 256 -> 3 (DenseWarp post-warp of ComposedWarp: dense translation field) */
#define LAB4D_NET_BG_BASE 6   /* background NeRF: posenc6 -> basefield (5+1 layers W=128, skip at 4) -> sdf head               */
#define LAB4D_NET_BG_COLOR 7  /* background NeRF: posenc8 -> colorfield (2+1, W=128) (+ feature) -> rgb (2) with the raw view
                                 direction as a second per-sample input (x2)                                              */
#define LAB4D_NET_SKIN18 8    /* the delta-skin net of the 18-joint skel-human skeleton: raw 54 bone coords -> 64 -> 64 -> 18 */
#define LAB4D_NET_HASH_GEO 9  /* hash-grid field (BASELINE config 5; no reference counterpart): raw 32 hash features -> 64 -> 16 (sdf + 15 geometry features) */
#define LAB4D_NET_HASH_COLOR 10 /* hash-grid field: raw [16 geometry features | 3 view direction] -> 64 -> 64 -> 3                                       */
#define LAB4D_NET_DENSE6 11   /* fg_motion "dense" (nnutils/warping.py:37-38,94-141): a bare DenseWarp with its class defaults, posenc6 -> CondMLP(D=6, W=256, skip at 4) -> 3 */
#define LAB4D_NET_SKIN_A 12   /* the delta-skin net with its FIRST LAYER IN PER-FRAME AFFINE FORM (emb_kind 2).  SkinningField.forward (nnutils/skinning.py:89-124)
                                 feeds the MLP the gaussian-scaled bone coordinates with PosEmbedding(3B, 0) = identity, and those are an affine
                                 function of the point per (frame, bone): linear_1 collapses to z0 = Wf[frame] [x; 1], Wf = W1[:, :3B] aff[frame]
                                 (+ the per-frame bias in the last column), a (64 x 4) table per frame formed by the caller.  The kernels
                                 evaluate relu(z0) as the net's "embedding" (fp32 FMAs, no 96-wide MFMA layer, no (S,3B) tensor in either
                                 direction) and run linear_2 / linear_final as layers 0 / 1: points (S,3) -> [64] -> 64 -> 25                 */
#define LAB4D_NET_SKIN18_A 13 /* the same for the 18-joint skeleton: points -> [64] -> 64 -> 18                                                  */
#define LAB4D_NET_COUNT 14

#define LAB4D_PREC_F32 0   /* v_mfma_f32_32x32x2_f32: exact fp32, parity path                       */
#define LAB4D_PREC_BF16 1  /* v_mfma_f32_32x32x16_bf16, fp32 accumulate: throughput path            */

typedef struct {
  int ke;      /* input columns taken from the embedded input (padded slot count, multiple of 32; 0 = none) */
  int kin;     /* input columns taken from the previous layer's activation                                  */
  int mout;    /* real output features                                                                       */
  int mout_pad;/* padded to a multiple of 32                                                                 */
  int relu;    /* ReLU after the affine map                                                                  */
  int pf_bias; /* layer takes a per-frame bias (M, mout_pad) in addition to its bias                         */
  int add_ext; /* after the activation, add the external feature tensor `ext` ([mout_pad][ld])           */
  int ext_grad;/* backward: an external gradient tensor is added to dL/d(output of this layer)               */
} lab4d_mlp_layer;

typedef struct {
  int n_layers;
  int emb_kind;   /* 0: posenc of a 3-vector with n_freq bands; 1: raw input with c_in channels;
                     2: slot j = relu(aff[frame][j] . [x; 1]) of a 3-vector x (per-frame affine first layer)   */
  int n_freq;
  int c_in;       /* raw input channels (emb_kind 1) or 3                                                   */
  int emb_slots;  /* real embedding slots: 6*n_freq+3 or c_in                                               */
  int ke;         /* padded embedding slots                                                                  */
  int c_out;      /* head channels                                                                            */
  lab4d_mlp_layer layers[LAB4D_MLP_MAX_LAYERS];
} lab4d_mlp_desc;

/* Embedding slot order of the posenc nets (what column j of the `ke` block means):
 *   slot 2*(3f+a)   = w_f * sin(2^f x_a)      slot 2*(3f+a)+1 = w_f * cos(2^f x_a)    f < n_freq, a < 3
 *   slot 6*n_freq+a = x_a ;  remaining slots are zero padding.
 * (the reference's channel order is [x, (f, {sin,cos}, a)], embedding.py:96-108; the host maps
 * weight columns with the col_map argument of lab4d_mlp_pack.) */
int lab4d_mlp_describe(int net, lab4d_mlp_desc* out);

/* bytes of the packed weight block of one layer (same for forward and transposed) */
int64_t lab4d_mlp_packed_bytes(int net, int layer, int precision);

/* Pack one layer's weight matrix into MFMA A-fragment order.
 *   W_ref: (mout, k_ref) fp32 row-major = the reference nn.Linear.weight (device pointer)
 *   col_map: (ke+kin) int32 device array: kernel input column -> column of W_ref, or -1 (zero)
 *   transposed = 0: forward operand (rows = outputs); 1: dgrad operand (rows = inputs) */
int lab4d_mlp_pack(int net, int layer, int precision, int transposed, const float* W_ref, int k_ref,
                   const int32_t* col_map, void* packed, void* stream);

typedef struct {
  int net, precision;
  int S;       /* samples                                                                                  */
  int S_pad;   /* samples rounded up to a multiple of 256 = one workgroup of 4 waves x 64 samples: the waves of a workgroup
                  run in lock-step (tail tiles are processed, never written to `out`)                             */
  int ld;      /* leading dimension (elements) of every [feature][sample] buffer: >= S_pad, multiple of 8.
                  Pad it so that ld*sizeof(store) is NOT a large power of two (e.g. S_pad*2 B + 4352 B):
                  a power-of-two row stride maps every feature row to the same HBM channel.             */
  int spf;     /* samples per frame (N*D); frame of sample s = s / spf                                     */
  const float* x;        /* (S,3) points or (S,c_in) raw inputs, fp32                                      */
  const float* freq_w;   /* (n_freq) annealing window weights or NULL (all ones)                           */
  const void* W[LAB4D_MLP_MAX_LAYERS];        /* packed forward weights                                    */
  const float* bias[LAB4D_MLP_MAX_LAYERS];    /* (mout_pad) fp32                                           */
  const float* pf_bias[LAB4D_MLP_MAX_LAYERS]; /* (M, mout_pad) fp32 per-frame bias of the pf_bias layers.  It REPLACES bias[l]
                                                 there (the host adds the shared bias into every row: the kernel fetches
                                                 one bias vector per tile, not two)                                   */
  void* act[LAB4D_MLP_MAX_LAYERS];  /* stored post-activation (blocked [64-sample block][mout_pad][64] + skew) or NULL.  Training mode (emb given): every
                                       hidden layer's buffer, or NONE (act[0] == NULL: the point-gradient-only mode of the sdf basefields LAB4D_NET_FG_BASE /
                                       _BG_BASE -- sign words + embedding stored, nothing else; no export buffer either).  Inference (emb NULL): only the
                                       layer another net consumes, if wanted */
  void* mask[LAB4D_MLP_MAX_LAYERS]; /* ReLU sign bits, uint32 [S_pad/TILE][mout_pad/32][64] (TILE = 64 bf16 / 32 fp32), or NULL */
  void* emb;                        /* [ke][ld] stored embedding or NULL                                */
  const void* ext;                  /* [mout_pad][ld] tensor added at the add_ext layer                 */
  float* out;                       /* (S, c_out) raw head output, fp32                                    */
  const float* x2;                  /* LAB4D_NET_BG_COLOR: (S,3) second per-sample input (view direction, nerf.py:196); else NULL */
  const int32_t* S_dev;             /* evaluation path only: device-side sample count -- only the first min(S, *S_dev) samples are
                                       processed (stream-compacted valid samples, nnutils/nerf.py:782-819, whose count stays on the
                                       device); NULL: all S                                                                       */
  const int32_t* frame_idx;         /* (S) int32 frame of every sample, or NULL: frame of sample s = s / spf                       */
  const float* aff;                 /* raw-input nets (LAB4D_NET_SKIN / _SKIN18) only, or NULL.  Non-NULL: x is the (S,3) POINTS and the net's
                                       c_in raw inputs are formed in the kernel as aff[frame][c][0..2] . x + aff[frame][c][3], aff = (M, c_in, 4)
                                       fp32 -- the gaussian-scaled bone coordinates of SkinningField.forward (skinning.py:126-140) in
                                       their per-frame affine form (lab4d_bone_affine), so the (S, 3B) tensor never exists in HBM.
                                       emb_kind 2 nets (LAB4D_NET_SKIN_A / _SKIN18_A): REQUIRED, (M, ke, 4) fp32 -- row j of frame m is
                                       [Wf[m][j][0..2] | bias]; x is the (S,3) points                                                   */
} lab4d_mlp_fwd_args;
int lab4d_mlp_forward(const lab4d_mlp_fwd_args* a, void* stream);

/* Fused backward of the NARROW networks (every layer <= 64 wide: LAB4D_NET_VIS, LAB4D_NET_SKIN_A, LAB4D_NET_SKIN18_A; bf16): autograd's dgrad
 * chain + one weight gradient per nn.Linear of VisField / SkinningField.delta_field (nnutils/visibility.py:39-63, skinning.py:70-124) in ONE
 * launch that needs NOTHING stored by the forward pass: it recomputes the forward of each 64-sample tile in registers (same arithmetic as
 * lab4d_mlp_forward), runs the dgrad chain, and accumulates every layer's dW = dZ X^T in registers across the tiles of a wave (operands
 * transposed through LDS, MFMA contraction over samples).  The forward of these nets therefore runs in inference mode (emb = act = mask =
 * NULL) also during training.  lab4d_mlp_fused_backward_supported(net, precision, spf) tells whether this entry applies (else: the stored-
 * activation path lab4d_mlp_backward + lab4d_mlp_wgrad).
 *   x (S,3) points; freq_w / aff / W / bias / pf_bias: exactly what the forward call took (W: forward operands of the hidden layers, WT:
 *   transposed operands of every layer); d_out (S, c_out) fp32.
 *   d_x (S,3) written, or NULL.  g_aff (M, 64, 4) accumulated (emb_kind 2 nets), or NULL.
 *   dW[l] (mout_pad, 64) fp32 in KERNEL column order (as lab4d_mlp_wgrad), db[l] (mout_pad), pf_db[l] (M, mout_pad) for the pf_bias layers
 *   (their db[l] is not written): all ACCUMULATED with atomics -- zero-fill first.  spf must be a multiple of 64 (a tile lies in one frame). */
typedef struct {
  int net, precision, S, spf;
  const float* x;
  const float* freq_w;
  const float* aff;
  const void* W[LAB4D_MLP_MAX_LAYERS];
  const void* WT[LAB4D_MLP_MAX_LAYERS];
  const float* bias[LAB4D_MLP_MAX_LAYERS];
  const float* pf_bias[LAB4D_MLP_MAX_LAYERS];
  const float* d_out;
  float* d_x;
  float* g_aff;
  float* dW[LAB4D_MLP_MAX_LAYERS];
  float* db[LAB4D_MLP_MAX_LAYERS];
  float* pf_db[LAB4D_MLP_MAX_LAYERS];
} lab4d_mlp_bwd_fused_args;
int lab4d_mlp_backward_fused(const lab4d_mlp_bwd_fused_args* a, void* stream);
int lab4d_mlp_fused_backward_supported(int net, int precision, int spf);

/* Tangent-mode forward of LAB4D_NET_FG_BASE for the eikonal term (nnutils/nerf.py:416-453, utils/torch_utils.py:4-27).
 * The SDF network is piecewise linear in its embedding e(x), so with v = d sdf / d e (the dgrad chain with d_out = 1),
 *   g = d sdf / d x = J_e(x)^T v        and        d L(g) / d theta = d/d theta [ u^T v(theta) ],  u = J_e(x) dL/dg.
 * u^T v is the output of the bias-free network with the primal's ReLU pattern applied to the input u, so its weight
 * gradient is dz(primal dgrad, d_out = 1) (x) t_{l-1}: run this entry point with x = u (S, ke) (embedding-slot order) and
 * mask[l] = the primal's sign bits, then lab4d_mlp_wgrad with the primal dz and the tangent act / emb stored here.
 * bias / pf_bias / ext are ignored. */
int lab4d_mlp_forward_tangent(const lab4d_mlp_fwd_args* a, void* stream);

typedef struct {
  int net, precision, S, S_pad, ld, spf;
  const void* WT[LAB4D_MLP_MAX_LAYERS];        /* packed transposed weights                                */
  const void* act[LAB4D_MLP_MAX_LAYERS];       /* stored post-activations from the forward (unused by the dgrad chain) */
  const void* mask[LAB4D_MLP_MAX_LAYERS];      /* ReLU sign bits written by the forward                    */
  const void* emb;                             /* stored embedding (posenc Jacobian)                       */
  const void* ext;                             /* unused (kept for ABI stability)                          */
  const float* d_out;                          /* (S, c_out) gradient of the head output                   */
  const void* ext_gin;                         /* [mout_pad][ld] gradient added at the ext_grad layer   */
  void* ext_gout;                              /* [mout_pad][ld] gradient wrt `ext` (written) or NULL   */
  void* dz[LAB4D_MLP_MAX_LAYERS];              /* [mout_pad][ld] dL/d(pre-activation) (written): every layer's, or NONE (dz[0] == NULL with d_x given:
                                                  point gradient only, the sdf basefields -- nothing for a weight gradient is written) */
  float* d_x;                                  /* (S,3) or (S,c_in) gradient wrt the input, or NULL        */
  float* d_x2;                                 /* LAB4D_NET_BG_COLOR: (S,3) gradient wrt x2 (written with d_x), or NULL */
  /* emb_kind 2 nets only (NULL otherwise): the adjoint of the per-frame affine first layer is taken inside the chain kernel */
  const float* x;                              /* (S,3) the points the forward saw                                     */
  const float* aff;                            /* (M, ke, 4) the table the forward saw                                  */
  float* g_aff;                                /* (M, ke, 4) dL/d aff, ACCUMULATED (atomicAdd; zero-fill first): row j of frame m =
                                                  sum over the frame's samples of dz0[s][j] * [x_s; 1] (reduced in registers per frame when
                                                  spf % 64 == 0, i.e. every tile lies in one frame; element-wise atomics otherwise);
                                                  d_x is then the (S,3) gradient wrt the points                                       */
} lab4d_mlp_bwd_args;
int lab4d_mlp_backward(const lab4d_mlp_bwd_args* a, void* stream);

/* Weight / bias gradients of one layer: dW[o][k] = sum_s dz[o][s] * X[k][s], db[o] = sum_s dz[o][s],
 * X = [emb (ke rows) ; act_prev (kin rows)].  dW: (mout_pad, ke+kin) fp32 row-major, db: (mout_pad);
 * both are ACCUMULATED into (atomicAdd) -- zero-fill first.  pf_db: (M, mout_pad) per-frame bias
 * gradient or NULL (accumulated, zero-fill first).  When pf_db is given, db is NOT written: db = sum_m pf_db[m]. */
int lab4d_mlp_wgrad(int net, int layer, int precision, int S, int S_pad, int ld, int spf, const void* dz, const void* emb,
                    const void* act_prev, float* dW, float* db, float* pf_db, int M, void* stream);
/* The same contraction with the result accumulated (atomically) straight into a gradient buffer in the REFERENCE layout --
 * what autograd would otherwise build from the kernel-layout matrix with a zero-fill, a column scatter and an add into
 * `weight.grad` (nnutils/base.py:65-78 layers; engine/trainer.py:343-350 accumulates them in .grad).  dW_ref: (mout, ld_ref)
 * fp32, typically the parameter's .grad; col_map: (ke + kin) int32, kernel input column -> reference column or -1 (the map
 * lab4d_mlp_pack takes); db: the reference bias gradient (mout entries, rows >= mout are not written) or NULL.
 * col_map == NULL is lab4d_mlp_wgrad. */
int lab4d_mlp_wgrad_mapped(int net, int layer, int precision, int S, int S_pad, int ld, int spf, const void* dz, const void* emb,
                           const void* act_prev, float* dW_ref, int ld_ref, const int32_t* col_map, float* db, float* pf_db, int M,
                           void* stream);

#endif /* LAB4D_MLP_H */
