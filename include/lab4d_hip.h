/*
 * lab4d_hip.h -- C ABI of the MI355X-native differentiable volume renderer (liblab4d_hip.so).
 *
 * This is the drop-in boundary for the Lab4D rendering hot path.  Every entry point takes
 * plain device pointers + sizes + a hipStream_t (passed as void*), launches hand-written
 * gfx950 kernels on that stream and returns 0 on success or a negative LAB4D_E* code
 * (lab4d_last_error() gives the message).  Nothing allocates, nothing retains pointers, the
 * caller owns all memory (reference convention: third_party/quaternion/quaternion.py:20-21).
 * All tensors are contiguous row-major.  M = frames, N = rays per frame, D = samples per ray,
 * S = M*N*D samples, B = bones.
 *
 * Each block cites the reference interface it replaces (paths relative to lab4d/).
 */
#ifndef LAB4D_HIP_H
#define LAB4D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LAB4D_OK 0
#define LAB4D_EINVAL (-1)   /* bad argument (null pointer, unsupported size/dtype) */
#define LAB4D_ELAUNCH (-2)  /* hipLaunch / runtime error */

/* dtype codes for the dqtorch-compatible entry points (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * third_party/quaternion/src/quaternion.cu:280-399) */
#define LAB4D_F32 0
#define LAB4D_F16 1
#define LAB4D_F64 2

const char* lab4d_last_error(void);
int lab4d_version(void);
/* returns the gfx arch string the library was compiled for ("gfx950") */
const char* lab4d_arch(void);
/* the kernel-experiment macros (LAB4D_ABL_*, LAB4D_WSABL_*, LAB4D_WS_TRACE, ... with the LAB4D_ prefix dropped, blank separated) the library was
 * compiled with; "" for the shipped build.  Most of them give WRONG results (timing ablations): callers refuse a library that reports any. */
const char* lab4d_build_flags(void);

/* ------------------------------------------------------------------------------------------
 * 1. dqtorch replacement -- third_party/quaternion/src/bindings.cpp:8-16, quaternion.h:11-24,
 *    matinv.h:7-11.  Same argument order as the pybind functions, raw pointers instead of
 *    at::Tensor, plus dtype and stream.  Launches on `stream` (the reference launches on the
 *    legacy default stream, quaternion.cu:240).
 * ------------------------------------------------------------------------------------------ */
/* out[b,:4] = in1[b] (x) in2[b]; D1,D2 in {3,4}: a 3-vector is a pure quaternion (quaternion.cu:29-64) */
int lab4d_quaternion_mul_forward(const void* in1, const void* in2, void* out, uint32_t B, uint32_t D1,
                                 uint32_t D2, int dtype, void* stream);
/* quaternion.cu:67-125 */
int lab4d_quaternion_mul_backward(const void* grad, uint32_t B, uint32_t D1, uint32_t D2, const void* in1,
                                  const void* in2, void* grad_in1, void* grad_in2, int dtype, void* stream);
/* quaternion.cu:128-199 (second order; needed by eikonal/normal through warps) */
int lab4d_quaternion_mul_backward_backward(const void* grad_out_1, const void* grad_out_2, uint32_t B,
                                           uint32_t D1, uint32_t D2, const void* grad, const void* in1,
                                           const void* in2, void* grad_grad, void* grad_grad_in1,
                                           void* grad_grad_in2, int dtype, void* stream);
/* quaternion.cu:202-217 */
int lab4d_quaternion_conjugate(const void* in, uint32_t B, void* out, int dtype, void* stream);
/* matinv.cu:42-62 / 80-116 / mat3x3_inv_forward / 119-240 */
int lab4d_mat3x3_det_forward(const void* in, void* out, uint32_t B, int dtype, void* stream);
int lab4d_mat3x3_scale_adjoint_forward(const void* in, const void* scales, void* out, uint32_t B, int dtype,
                                       void* stream);
int lab4d_mat3x3_inv_forward(const void* in, void* out, void* out_scales, uint32_t B, int dtype, void* stream);
int lab4d_mat3x3_inv_backward(const void* grad, const void* inv_mats, void* grad_in, uint32_t B, int dtype,
                              void* stream);

/* ------------------------------------------------------------------------------------------
 * 2. Ray sampling -- utils/render_utils.py:8-56 (sample_cam_rays, perturb=False) fused with
 *    nnutils/nerf.py:821-844 (NeRF.cam_to_field).  fp32.
 *    hxy (M,N,3), Kinv (M,3,3), near_far (M,2); depth_in (M,N,D) or NULL (uniform in [near,far]).
 *    cam2field_q (M,4)/cam2field_t (M,3) or NULL: if given also emits field-space points.
 *    Outputs (any may be NULL): xyz_cam,dir_cam (S,3), deltas,depth (S), xyz_field,dir_field (S,3).
 * ------------------------------------------------------------------------------------------ */
int lab4d_ray_samples_forward(const float* hxy, const float* Kinv, const float* near_far, const float* depth_in,
                              const float* cam2field_q, const float* cam2field_t, int M, int N, int D,
                              float* xyz_cam, float* dir_cam, float* deltas, float* depth, float* xyz_field,
                              float* dir_field, void* stream);
/* Adjoint: accumulates (atomicAdd) into g_Kinv (M,9), g_q (M,4), g_t (M,3); caller zero-fills.
 * Any g_* input may be NULL (treated as zero). */
int lab4d_ray_samples_backward(const float* hxy, const float* Kinv, const float* near_far, const float* depth_in,
                               const float* cam2field_q, const float* cam2field_t, int M, int N, int D,
                               const float* g_xyz_cam, const float* g_dir_cam, const float* g_deltas,
                               const float* g_xyz_field, const float* g_dir_field, float* g_Kinv, float* g_q,
                               float* g_t, void* stream);

/* utils/render_utils.py:187-233 sample_pdf(det=True): bins (R,n_samples+1... see below), weights (R,n_w).
 * bins has n_w+1 entries per ray.  Writes samples (R,n_imp) and the searchsorted(right=True) indices
 * inds (R,n_imp) int64 -- bit-exact vs torch (sequential fp32 cumsum). */
int lab4d_sample_pdf(const float* bins, const float* weights, int R, int n_w, int n_imp, float eps,
                     float* samples, int64_t* inds, void* stream);
/* utils/render_utils.py:209-213 sample_pdf(det=False): the same inverse-CDF sampling at caller-drawn uniforms u_sorted (R,n_imp),
 * ASCENDING per ray (the caller draws torch.rand like the reference, sorts per ray and un-sorts samples / inds afterwards:
 * searchsorted acts per element).  u_sorted == NULL is lab4d_sample_pdf. */
int lab4d_sample_pdf_u(const float* bins, const float* weights, const float* u_sorted, int R, int n_w, int n_imp, float eps,
                       float* samples, int64_t* inds, void* stream);
/* nnutils/nerf.py:731: sort(cat([depth_a, depth_b], -1)) per ray; a,b: (R,na),(R,nb) -> out (R,na+nb). */
int lab4d_sort_depth(const float* a, int na, const float* b, int nb, int R, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * 2b. Valid-sample compaction of the evaluation path -- nnutils/nerf.py:495-528 (get_valid_idx), 769-819 (query_nerf);
 *     utils/geom_utils.py:409-422 (extend_aabb), 506-517 (check_inside_aabb, strict inequalities).
 *     valid_mask: mask[s] = xyz[s] strictly inside aabb (2,3) AND (t_aabb == NULL or xyz_t[s] strictly inside t_aabb (2,3)); uint8.
 *     compact: idx[0..count) = ascending indices of the set mask bytes, *count = how many -- BOTH stay on the device (the
 *       reference's `valid_idx.sum()` and boolean indexing each cost a host round trip); work: lab4d_compact_work_ints(S) ints.
 *     gather_rows / scatter_rows / frame_of take that device-side count: rows j < *count are moved, gather zero-fills the rest
 *       of its n_rows-row output, scatter leaves the other rows of dst as the caller filled them (zeros, nerf.py:812-816),
 *       frame_of writes idx[j] / spf (the frame a compacted sample belongs to: lab4d_mlp_fwd_args.frame_idx).
 * ------------------------------------------------------------------------------------------ */
int lab4d_valid_mask(const float* xyz, const float* xyz_t, const float* aabb, const float* t_aabb, long S, unsigned char* mask, void* stream);
long lab4d_compact_work_ints(long S);
int lab4d_compact(const unsigned char* mask, long S, int* idx, int* count, int* work, void* stream);
int lab4d_gather_rows(const float* src, const int* idx, const int* count, long n_rows, int C, float* dst, void* stream);
int lab4d_scatter_rows(const float* src, const int* idx, const int* count, long n_rows, int C, float* dst, void* stream);
int lab4d_frame_of(const int* idx, const int* count, long n_rows, int spf, int* frame, void* stream);

/* ------------------------------------------------------------------------------------------
 * 3. Compositing -- utils/render_utils.py:59-184 (render_pixel / compute_weights / integrate).
 *    One launch renders every per-sample field of a ray batch.  R = M*N rays, D samples.
 *    fields[i]: (R,D,channels[i]) fp32; modes[i]: 0 = normalised weights w/(sum w+1e-6),
 *    1 = same weights but detached (key_freeze, render_utils.py:147), 2 = plain mean over D
 *    (eikonal / delta_skin, render_utils.py:75-80).
 * ------------------------------------------------------------------------------------------ */
#define LAB4D_MAX_FIELDS 16
typedef struct {
  int n_fields;
  const float* fields[LAB4D_MAX_FIELDS];
  int channels[LAB4D_MAX_FIELDS];
  int modes[LAB4D_MAX_FIELDS];
} lab4d_field_list;
typedef struct {
  int n_fields;
  float* fields[LAB4D_MAX_FIELDS];
} lab4d_field_grads;

/* density, deltas: (R,D).  Optional (NULL to skip): flow (R,D,3) = (u,v,valid), vis (R,D) logits,
 * gauss_density (R,D).  Outputs: weights,transmit (R,D) (may be NULL), mask (R), out (R,sum channels)
 * in field order, flow_out (R,2), vis_num (R) = -mean_D(logsigmoid(vis)*T), t_sum (R) = sum_D T,
 * gauss_mask (R). */
int lab4d_composite_forward(const float* density, const float* deltas, const lab4d_field_list* fl,
                            const float* flow, const float* vis, const float* gauss_density, int R, int D,
                            float* weights, float* transmit, float* mask, float* out, float* flow_out,
                            float* vis_num, float* t_sum, float* gauss_mask, void* stream);
/* Adjoint of the above.  g_* inputs: g_mask (R), g_out (R,sumC), g_flow_out (R,2), g_vis_num (R),
 * g_gauss_mask (R); each may be NULL.  Outputs (each may be NULL): g_density, g_deltas (R,D),
 * g_fields[i] (R,D,c_i), g_flow (R,D,3; valid channel gets 0), g_vis (R,D), g_gauss_density (R,D). */
int lab4d_composite_backward(const float* density, const float* deltas, const lab4d_field_list* fl,
                             const float* flow, const float* vis, const float* gauss_density, int R, int D,
                             const float* g_mask, const float* g_out, const float* g_flow_out,
                             const float* g_vis_num, const float* g_gauss_mask, float* g_density,
                             float* g_deltas, const lab4d_field_grads* g_fields, float* g_flow, float* g_vis,
                             float* g_gauss_density, void* stream);

/* ------------------------------------------------------------------------------------------
 * 3a. Per-sample epilogues of the training query.
 *   flow_cyc -- NeRF.compute_flow (nnutils/nerf.py:948-997: field_to_cam with the pair partner's pose, utils/geom_utils.py:14-27
 *     pinhole_projection, flow = projection - pixel, valid = z > 1e-6 [and |flow| < flow_thresh; pass a negative thresh for None])
 *     and Deformable.cycle_loss' distance (nnutils/deformable.py:189-193), one kernel each way.
 *     xyz_next (S,3): canonical samples forward-warped into the partner's object space; q (M,4), t (M,3): the partner's
 *     field-to-camera pose (row m serves samples [m*spf, (m+1)*spf)); K (M,3,3): its intrinsics (Kmatinv(Kinv)); hxy (S/D,3): pixel of
 *     every ray; xyz_cyc / xyz_t (S,3) or NULL: the forward-warped and the time-t points of the cycle term.
 *     -> flow (S,3) = (u, v, valid), cyc (S) = |xyz_cyc - xyz_t|.
 *     backward: g_flow (S,3; the valid channel is ignored), g_cyc (S) -> g_xyz_next (S,3), g_per_frame (M,16) = [g_q 4 | g_t 3 | g_K 9]
 *     (zero-filled by the call), g_xyz_cyc, g_xyz_t (S,3) or NULL.
 *   volsdf -- the VolSDF density of NeRF.forward (nerf.py:186-192): density = (0.5 + 0.5 sign(sdf) expm1(-|sdf| ibeta)) ibeta with
 *     ibeta a 1-element device tensor; backward gives g_sdf (S) and g_ibeta (1, zero-filled by the call; may be NULL).
 * ------------------------------------------------------------------------------------------ */
int lab4d_flow_cyc_forward(const float* xyz_next, const float* q, const float* t, const float* K, const float* hxy, const float* xyz_cyc,
                           const float* xyz_t, long S, int spf, int D, float flow_thresh, float* flow, float* cyc, void* stream);
int lab4d_flow_cyc_backward(const float* xyz_next, const float* q, const float* t, const float* K, const float* xyz_cyc, const float* xyz_t,
                            const float* g_flow, const float* g_cyc, long S, int spf, int M, float* g_xyz_next, float* g_per_frame,
                            float* g_xyz_cyc, float* g_xyz_t, void* stream);
int lab4d_volsdf_forward(const float* sdf, const float* ibeta, long S, float* density, void* stream);
int lab4d_volsdf_backward(const float* sdf, const float* ibeta, const float* g_density, long S, float* g_sdf, float* g_ibeta, void* stream);

/* ------------------------------------------------------------------------------------------
 * 3b. Field composition -- nnutils/multifields.py:339-398 (MultiFields.compose_fields, "comp" configs).
 *    The samples of two fields (fg: Da per ray, bg: Db per ray) are concatenated along the depth axis, z-sorted
 *    (argsort of the concatenated depth) and every per-sample key is gathered with that permutation.
 *    compose_order: depth_a (R,Da), depth_b (R,Db) -> order (R,Da+Db) int32 = argsort (stable: ties keep the
 *    concatenation order) and pos (R,Da+Db) = its inverse (where each input sample went; the adjoint's gather index).
 *    compose_gather: out (R,Dn,C) = [a (R,Da,C) | b (R,Db,C)][idx (R,Dn), row stride idx_ld]; a NULL part reads as
 *    zeros (a key only one field produces, multifields.py:383-389).  Forward: idx = order, Dn = Da+Db.  Adjoint of part a:
 *    a = g_out with Da := Da+Db, b = NULL, idx = pos, Dn = Da (part b: idx = pos + Da, Dn = Db).
 * ------------------------------------------------------------------------------------------ */
int lab4d_compose_order(const float* depth_a, int Da, const float* depth_b, int Db, int R, int32_t* order, int32_t* pos, void* stream);
int lab4d_compose_gather(const float* a, int Da, const float* b, int Db, const int32_t* idx, int idx_ld, int R, int Dn, int C,
                         float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * 3c. Per-sample epilogues -- nnutils/feature.py:149-150 (FeatureNeRF.compute_feat): y = x / ||x||_2 over the last axis
 *     (C = 16 feature channels, or 3), no epsilon, exactly the reference expression; backward g_x = (g - y (y.g)) / ||x||.
 * ------------------------------------------------------------------------------------------ */
int lab4d_l2_normalize_forward(const float* x, int S, int C, float* y, void* stream);
int lab4d_l2_normalize_backward(const float* x, const float* g, int S, int C, float* g_x, void* stream);

/* Eikonal term, backward (nnutils/nerf.py:416-453; see lab4d_mlp_forward_tangent in lab4d_mlp.h): from the points x (S,3), the sdf gradient g (S,3)
 * and the incoming gradient ge (S) of (|g| - 1)^2, the tangent-mode input u (S, ke) = J_e(x) dL/dg in embedding-slot order (n_freq bands, optional
 * annealing weights freq_w), zero-padded to ke columns. */
int lab4d_eikonal_tangent_input(const float* x, const float* g, const float* ge, const float* freq_w, int S, int n_freq, int ke, float* u,
                                void* stream);

/* ------------------------------------------------------------------------------------------
 * 3d. FeatureNeRF.global_match -- nnutils/feature.py:152-199: soft arg-max of the (R,16) pixel features over K <= 1024 canonical
 *     candidates (the caller draws them: torch.randperm, feature.py:176-178, and gathers their rows),
 *        score = (feat_px @ feat_c^T) * exp(logsigma);  prob = softmax(score, dim=1);  xyz_matched = prob @ xyz_c.
 *     forward: out (R,3); stats (R,2) = per-row (max score, sum exp(score - max)), handed back to backward.
 *     backward: g_feat_c (K,16), g_xyz_c (K,3), g_logsigma (1) are WRITTEN; the pixel features are data (no gradient, as in
 *     the reference: samples_dict["feature"]).  work: lab4d_global_match_workspace_floats(K) floats.  The (R,K) score / prob
 *     matrices never exist in memory.  logsigma: device scalar (nn.Parameter, feature.py:86-87).
 * ------------------------------------------------------------------------------------------ */
int lab4d_global_match_workspace_floats(int K);
int lab4d_global_match_forward(const float* feat_px, const float* feat_c, const float* xyz_c, const float* logsigma, int R, int C, int K,
                               float* out, float* stats, void* stream);
int lab4d_global_match_backward(const float* feat_px, const float* feat_c, const float* xyz_c, const float* logsigma, const float* out,
                                const float* stats, const float* g_out, int R, int C, int K, float* g_feat_c, float* g_xyz_c,
                                float* g_logsigma, float* work, void* stream);

/* ------------------------------------------------------------------------------------------
 * 4. Fused positional-encoding + MLP stack -- nnutils/embedding.py:69-125 (PosEmbedding),
 *    nnutils/base.py:65-78,123-150 (BaseMLP/CondMLP), as used by nnutils/nerf.py:167-215,
 *    visibility.py:53-63, feature.py:136-150, skinning.py:108-119, warping.py:143-170.
 *    See lab4d_mlp.h for the network descriptor.
 * ------------------------------------------------------------------------------------------ */
#include "lab4d_mlp.h"

/* ------------------------------------------------------------------------------------------
 * 5. Linear-blend skinning with dual quaternions -- nnutils/warping.py:277-336,
 *    skinning.py:89-153, utils/transforms.py:9-25, utils/geom_utils.py:45-83,
 *    utils/loss_utils.py:21-42.  See lab4d_skin.h.
 * ------------------------------------------------------------------------------------------ */
#include "lab4d_skin.h"

/* ------------------------------------------------------------------------------------------
 * 6. Skeleton forward kinematics of the per-frame articulation path --
 *    utils/skel_utils.py:50-145, utils/geom_utils.py:110-140, utils/quat_transform.py:468-532,
 *    nnutils/pose.py:417-502.  See lab4d_pose.h.
 * ------------------------------------------------------------------------------------------ */
#include "lab4d_pose.h"

/* ------------------------------------------------------------------------------------------
 * 7. Optimizer step -- engine/trainer.py:164-190,350,581-604.  See lab4d_optim.h.
 * ------------------------------------------------------------------------------------------ */
#include "lab4d_optim.h"

/* ------------------------------------------------------------------------------------------
 * 8. Multiresolution hash encoding (BASELINE config 5; not in the reference, parity unpinned).  See lab4d_hashgrid.h.
 * ------------------------------------------------------------------------------------------ */
#include "lab4d_hashgrid.h"

/* ------------------------------------------------------------------------------------------
 * 9. Batch ingestion -- dataloader/vidloader.py:217-358, utils/numpy_utils.py:97-122.  See lab4d_ingest.h.
 * ------------------------------------------------------------------------------------------ */
#include "lab4d_ingest.h"

/* ------------------------------------------------------------------------------------------
 * 10. The per-frame MLPs of the pose / appearance path as one launch each way -- nnutils/embedding.py:177-217,
 *     nnutils/base.py:65-78, nnutils/time.py:65-73, nnutils/pose.py:103-147,442-447, nnutils/intrinsics.py:73-107.  See lab4d_rowmlp.h.
 * ------------------------------------------------------------------------------------------ */
#include "lab4d_rowmlp.h"

#ifdef __cplusplus
}
#endif
#endif /* LAB4D_HIP_H */
