/*
 * lab4d_rowmlp.h -- the per-frame (M-row) MLPs in front of the hot path as ONE launch each way (included by lab4d_hip.h).
 * SURVEY.md 8f row 1, the part round 5 left on torch modules: TimeEmbedding, TimeMLP and the heads of CameraMLP / IntrinsicsMLP /
 * ArticulationSkelMLP / ArticulationFlatMLP / AppearanceEmbedding.
 *
 * Replaces (paths relative to lab4d/):
 *   nnutils/embedding.py:177-217   TimeEmbedding.frame_to_tid + forward: PosEmbedding(1, F)(t), mapping1, cat with the video's
 *                                  InstEmbedding row, mapping2
 *   nnutils/embedding.py:69-125    PosEmbedding.forward for one input channel (no annealing: TimeEmbedding never sets alpha)
 *   nnutils/base.py:65-78          BaseMLP.forward: D x (Linear + ReLU) + linear_final (+ ReLU), skips = []
 *   nnutils/pose.py:103-147        CameraMLP.forward's two heads (nn.Sequential(Linear, ReLU, Linear))
 *   nnutils/intrinsics.py:73-84    IntrinsicsMLP.forward's focal head
 *   nnutils/pose.py:442-447        ArticulationSkelMLP's so3 head; nnutils/appearance.py:46-56
 * The reference runs each nn.Linear / ReLU / cat / index as its own launch on M <= a few hundred rows (~40 launches forward and
 * ~100 backward per module and step); here a module's whole chain is a PROGRAM of dense layers over a per-row strip of floats:
 * one launch forward, two backward (input-gradient chain; every weight / bias / embedding-row gradient).
 *
 * Data model.  `work` is (M, row_stride) fp32, row-major: a row's strip holds every intermediate of that row.  A layer reads
 * columns [src_col, src_col + in_dim) and writes act(x W^T + b) to columns [dst_col, dst_col + out_dim); a concatenation is two
 * producers writing adjacent column ranges; a fan-out (one feature, two heads) is two layers naming the same source.  Layers run in
 * index order; the column ranges a layer writes must not overlap anything an earlier or later layer of the program writes.
 * The optional time prologue fills the Fourier columns and the instance-code columns from frame ids before layer 0.
 * Backward: `gwork` (M, row_stride) holds dL/d(strip): the chain kernel clears it, copies the output gradients in (out[i]), turns every
 * layer's output gradient into dZ in place (ReLU mask from `work`) and ADDS dZ W into the source columns; the parameter kernel forms
 * dW = dZ^T X, db = colsum(dZ) per layer and the instance-embedding rows' gradient with ONE owner per element (deterministic, no atomics) --
 * written, or added to the buffer it is given (the `acc` bits: straight into `weight.grad`, no AccumulateGrad launch per parameter).
 *
 * All fp32 FMA arithmetic (the precision the reference computes these modules in); W in the reference's (out, in) layout: no
 * packing, checkpoint-compatible.  Limits: n_layers <= 16, every in_dim / out_dim <= 1024, n_freq <= 16.
 */
#ifndef LAB4D_ROWMLP_H
#define LAB4D_ROWMLP_H

#define LAB4D_ROWMLP_MAX_LAYERS 16

typedef struct {
  const float* W; /* (out_dim, in_dim) row-major = nn.Linear.weight */
  const float* b; /* (out_dim) or NULL */
  float* dW;      /* backward: (out_dim, in_dim) written -- or, with acc bit 0, ADDED to (e.g. weight.grad: one owner per element, no atomics); NULL: not wanted */
  float* db;      /* backward: (out_dim) written / added to (acc bit 1); NULL: not wanted */
  int32_t in_dim, out_dim, src_col, dst_col;
  int32_t relu;   /* 1: ReLU on the output */
  int32_t acc;    /* backward: bit 0: dW is accumulated into, bit 1: db is accumulated into */
} lab4d_rowmlp_layer;

/* A column range of the strip bound to a caller tensor (M, width), contiguous fp32.
 * in[i]:  forward: copied INTO the strip before layer 0 (an external input, e.g. a time embedding computed elsewhere);
 *         backward: the gradient of that input is written there (ptr NULL: not wanted).
 * out[i]: forward: the columns are written there behind the last layer (ptr NULL: nothing);
 *         backward: the gradient of that output is read from there (ptr NULL: zero). */
typedef struct {
  float* ptr;
  int32_t col, width;
} lab4d_rowmlp_io;
#define LAB4D_ROWMLP_MAX_IO 4

typedef struct {
  int32_t n_layers, row_stride;
  /* time prologue (embedding.py:177-217), enabled by frame_id != NULL: row m takes frame f = frame_id[m] (raw frame id),
   * t = ((f - vstart[f]) - vidlen[f] / 2) / max_ts * 2 * time_scale, writes [t, sin(2^k t), cos(2^k t)]_{k < n_freq} to columns
   * [four_col, four_col + 2 n_freq + 1) and row (inst_rows == 1 ? 0 : vid[f]) of inst_W (inst_rows, inst_dim) to columns
   * [inst_col, inst_col + inst_dim). */
  const int64_t* frame_id; /* (M) */
  const int64_t* vstart;   /* raw_fid_to_vstart (N) */
  const int64_t* vidlen;   /* raw_fid_to_vidlen (N) */
  const int64_t* vid;      /* raw_fid_to_vid (N) */
  const float* inst_W;     /* InstEmbedding.mapping.weight (inst_rows, inst_dim) */
  float* d_inst_W;         /* backward: (inst_rows, inst_dim) written; NULL: not wanted */
  float max_ts, time_scale;
  int32_t n_freq, four_col, inst_col, inst_dim, inst_rows;
  int32_t acc_inst;        /* backward: 1: d_inst_W is accumulated into */
  int32_t n_in, n_out;
  lab4d_rowmlp_io in[LAB4D_ROWMLP_MAX_IO], out[LAB4D_ROWMLP_MAX_IO];
  lab4d_rowmlp_layer layer[LAB4D_ROWMLP_MAX_LAYERS];
} lab4d_rowmlp_prog;

/* Runs the prologue, the inputs' copy-in, every layer of `prog` (a HOST struct, copied into the launch) and the outputs' copy-out on rows [0, M) of
 * work (M, row_stride), which afterwards holds every intermediate (the backward needs it). */
int lab4d_rowmlp_forward(const lab4d_rowmlp_prog* prog, float* work, int M, void* stream);
/* Two launches: (1) gwork (M, row_stride; need NOT be initialised) is cleared, the outputs' gradients are copied in, the dZ / input-gradient chain runs
 * in place, the inputs' gradients are copied out; (2) every layer's dW / db and d_inst_W. */
int lab4d_rowmlp_backward(const lab4d_rowmlp_prog* prog, const float* work, float* gwork, int M, void* stream);

/* The element-wise epilogues behind the heads, one launch each way (the reference: ~12 launches forward, ~25 backward each).
 * A row's video = V == 1 ? 0 : vid[frame_id[m]] (frame_id / vid may be NULL when V == 1).
 *
 * CameraMLP.get_vals (nnutils/pose.py:126-147): out (M,4) = quaternion_mul(F.normalize(raw (M,4)), F.normalize(base_quat (V,4)[video])).
 * Backward: g_raw (M,4) written; g_base (V,4) written or (acc_base) added to, or NULL; row_scratch: (M,4) floats of scratch. */
int lab4d_camera_epilogue_forward(const float* raw, const float* base_quat, const int64_t* frame_id, const int64_t* vid, int M, int V, float* out,
                                  void* stream);
int lab4d_camera_epilogue_backward(const float* raw, const float* base_quat, const int64_t* frame_id, const int64_t* vid, const float* g_out, int M, int V,
                                   float* g_raw, float* row_scratch, float* g_base, int acc_base, void* stream);
/* IntrinsicsMLP.get_vals (nnutils/intrinsics.py:94-107): f = exp(raw (M,2)) * exp(base_logfocal (V,2)[video]); out (M,4) = [mean(f), mean(f),
 * base_ppoint (V,2)[video]].  Backward: g_raw (M,2) written; g_video (V,4) = [d base_logfocal | d base_ppoint] written, or NULL. */
int lab4d_intrinsics_epilogue_forward(const float* raw, const float* base_logfocal, const float* base_ppoint, const int64_t* frame_id, const int64_t* vid,
                                      int M, int V, float* out, void* stream);
int lab4d_intrinsics_epilogue_backward(const float* raw, const float* base_logfocal, const int64_t* frame_id, const int64_t* vid, const float* g_out, int M,
                                       int V, float* g_raw, float* row_scratch, float* g_video, void* stream);

#endif /* LAB4D_ROWMLP_H */
