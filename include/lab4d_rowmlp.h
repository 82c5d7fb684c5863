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
 * Backward: `gwork` (M, row_stride) holds dL/d(strip); the caller zero-fills it and writes the gradients of the columns it consumed;
 * the chain kernel turns every layer's output gradient into dZ in place (ReLU mask from `work`) and ADDS dZ W into the source
 * columns; the parameter kernel forms dW = dZ^T X, db = colsum(dZ) per layer (written, not accumulated: deterministic, no atomics)
 * and the instance-embedding rows' gradient.
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
  float* dW;      /* backward: (out_dim, in_dim) written; NULL: not wanted */
  float* db;      /* backward: (out_dim) written; NULL: not wanted */
  int32_t in_dim, out_dim, src_col, dst_col;
  int32_t relu;   /* 1: ReLU on the output */
  int32_t pad_;
} lab4d_rowmlp_layer;

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
  int32_t n_freq, four_col, inst_col, inst_dim, inst_rows, pad_;
  lab4d_rowmlp_layer layer[LAB4D_ROWMLP_MAX_LAYERS];
} lab4d_rowmlp_prog;

/* Runs the prologue and every layer of `prog` (a HOST struct, copied into the launch) on rows [0, M) of work. */
int lab4d_rowmlp_forward(const lab4d_rowmlp_prog* prog, float* work, int M, void* stream);
/* Two launches: the dZ / input-gradient chain over gwork (in place), then every layer's dW / db and d_inst_W. */
int lab4d_rowmlp_backward(const lab4d_rowmlp_prog* prog, const float* work, float* gwork, int M, void* stream);

#endif /* LAB4D_ROWMLP_H */
