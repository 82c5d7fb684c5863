/*
 * lab4d_skin.h -- linear-blend skinning with dual quaternions (included by lab4d_hip.h).
 *
 * Replaces (paths relative to lab4d/):
 *   utils/transforms.py:9-25        get_bone_coords            -> lab4d_bone_coords_*
 *   nnutils/skinning.py:89-153      SkinningField.forward      (bone coords / gauss, dist2, delta = relu(mlp)*0.1)
 *   nnutils/warping.py:316-333      softmax, skin_entropy, delta_skin
 *   utils/geom_utils.py:45-83       dual_quaternion_skinning   (hemisphere fix, blend, normalise, apply)
 *   utils/loss_utils.py:21-42       cross_entropy_skin_loss    -> lab4d_skin_blend_*
 * The reference materialises ~10 tensors of shape (M,N,D,B,4) (210 MB each at 524k samples); here
 * every per-(sample,bone) quantity lives in registers.  The delta-skinning MLP between the two
 * stages is LAB4D_NET_SKIN of lab4d_mlp.h.  All fp32.  S samples, frame of sample s = s / spf,
 * M frames, B bones (25).
 */
#ifndef LAB4D_SKIN_H
#define LAB4D_SKIN_H

/* xyz_bone[s][b][:] = apply(inverse(bone2obj[frame(s)][b]), xyz[s]) / gauss[b]   -> (S, 3B)
 * art_r, art_d: (M,B,4) real / dual parts of the bone-to-object dual quaternions; gauss: (B,3). */
int lab4d_bone_coords_forward(const float* xyz, const float* art_r, const float* art_d, const float* gauss, int S,
                              int spf, int M, int B, float* xyz_bone, void* stream);
/* g_bone (S,3B) -> g_xyz (S,3) written; g_art_r, g_art_d (M,B,4), g_gauss (B,3) accumulated (zero-fill first).
 * The three parameter gradients may be NULL: the host then derives them from lab4d_gram_per_frame(g_bone, [xyz,1])
 * (the bone transform is affine in the point), which reads g_bone once instead of once per bone. */
int lab4d_bone_coords_backward(const float* xyz, const float* art_r, const float* art_d, const float* gauss,
                               const float* g_bone, int S, int spf, int M, int B, float* g_xyz, float* g_art_r,
                               float* g_art_d, float* g_gauss, void* stream);

/* lab4d_bone_coords_backward's point gradient AND the per-frame Gram matrix of the parameter path in one pass over the (S,3B) gradient:
 * g_xyz (S,3) is written, G (M,3B,4) = sum over the frame's samples of g_bone[s,:]^T [x_s, 1] is zero-filled and accumulated (feed it to
 * lab4d_bone_params_from_gram).  Needs spf % 256 == 0 (every 256-sample tile inside one frame); otherwise use lab4d_bone_coords_backward
 * + lab4d_gram_per_frame. */
int lab4d_bone_coords_backward_gram(const float* xyz, const float* art_r, const float* gauss, const float* g_bone, int S, int spf, int M, int B,
                                    float* g_xyz, float* G, void* stream);

/* Parameter gradients of lab4d_bone_coords_forward from the per-frame Gram matrix G (M,B,3,4) =
 * lab4d_gram_per_frame(g_bone (S,3B), [xyz,1] (S,4)): g_art_r, g_art_d (M,B,4) written, g_gauss (B,3) accumulated
 * (zero-fill first); any of the three may be NULL.  Replaces the reference's autograd through
 * transforms.get_bone_coords / quaternion ops (lab4d/nnutils/skinning.py:126-140) for the (M,B)-sized parameters. */
int lab4d_bone_params_from_gram(const float* art_r, const float* art_d, const float* gauss, const float* G, int M, int B,
                                float* g_art_r, float* g_art_d, float* g_gauss, void* stream);

/* skin = -(|c_b|^2 + relu(delta_raw_b)*0.1) with c_b the gaussian-scaled bone coordinate of the point (recomputed from
 * art_r, art_d (M,B,4), gauss (B,3): the same arithmetic as lab4d_bone_coords_forward, so the (S,3B) tensor is only ever
 * read by the delta-skin MLP); p = softmax(skin); blend the per-bone transforms se3 (M,B,4)x2 with the arg-max bone's
 * hemisphere; out = apply(blend, xyz).
 * Outputs: out (S,3), entropy (S) = logsumexp(skin) - max(skin), dskin (S) = mean_b delta_b^2.  `work`: scratch of
 * M*B*12 floats (the per-frame affine form of the bone coordinates). */
int lab4d_skin_blend_forward(const float* xyz, const float* art_r, const float* art_d, const float* gauss,
                             const float* delta_raw, const float* se3_r, const float* se3_d, int S, int spf, int M, int B,
                             float* out, float* entropy, float* dskin, float* work, void* stream);
/* Adjoint.  g_ent / g_dskin may be NULL.  Writes g_xyz (S,3) (including the path through the bone coordinates) and
 * g_raw (S,B); accumulates g_se3 (M,B,8) = [d/d se3_r (4) | d/d se3_d (4)] (zero-fill first); writes g_art_r, g_art_d
 * (M,B,4) and accumulates g_gauss (B,3) (zero-fill first) -- the three may be NULL together.  The bone-coordinate
 * gradient dL/dc_b = -2 dL/dskin_b * c_b is never materialised: it reaches the point analytically and the parameters
 * through per-frame second moments.  `work`: scratch of M*B*34 floats when spf % 256 == 0 and g_se3 is given (every 256-sample
 * tile then lies in one frame and the per-frame reductions are formed inside the kernel), else M*B*34 + S*(2B+18) floats. */
int lab4d_skin_blend_backward(const float* xyz, const float* art_r, const float* art_d, const float* gauss,
                              const float* delta_raw, const float* se3_r, const float* se3_d, const float* g_out,
                              const float* g_ent, const float* g_dskin, int S, int spf, int M, int B, float* g_xyz,
                              float* g_raw, float* g_se3, float* g_art_r, float* g_art_d, float* g_gauss, float* work,
                              void* stream);
/* The same with accumulate != 0: g_xyz (S,3) and g_raw (S,B) are ADDED to instead of written (they hold the adjoint of an earlier blend of the
 * same skinning field evaluation: the training graph warps every canonical sample forward twice off ONE delta-skin evaluation, nerf.py:966-973
 * and deformable.py:173-198); g_se3 / g_art_* / g_gauss are written as above. */
int lab4d_skin_blend_backward_acc(const float* xyz, const float* art_r, const float* art_d, const float* gauss,
                                  const float* delta_raw, const float* se3_r, const float* se3_d, const float* g_out,
                                  const float* g_ent, const float* g_dskin, int S, int spf, int M, int B, float* g_xyz,
                                  float* g_raw, float* g_se3, float* g_art_r, float* g_art_d, float* g_gauss, float* work,
                                  int accumulate, void* stream);

/* Size of `work` (floats) for lab4d_skin_blend_backward[_acc] with these arguments -- the library decides between the fused and the unfused
 * adjoint (spf, g_se3 given, the LAB4D_BLEND_FUSE experiment switch) in ONE place, and the caller sizes the scratch by asking: M*B*34 for the
 * fused path, M*B*34 + S*(2B+18) otherwise.  Host-only, no launch. */
long long lab4d_skin_blend_backward_workspace_floats(int S, int spf, int M, int B, int has_g_se3);

/* Gaussian-bone density  max_b exp(-0.5 |x - c_b|^2 / 0.01^2) * ibeta  (nnutils/deformable.py:329-356,
 * warping.py:355-387, utils/transforms.py:28-40).  centres: (B,3); ibeta: device scalar (no host sync).
 * best: (S) int32 arg-min bone (saved for
 * the adjoint).  Backward writes g_xyz (S,3) and accumulates g_centres (B,3), g_ibeta (1). */
int lab4d_gauss_density_forward(const float* xyz, const float* centres, int B, const float* ibeta, int S, float* out,
                                int* best, void* stream);
int lab4d_gauss_density_backward(const float* xyz, const float* centres, int B, const float* ibeta, const int* best,
                                 const float* g, int S, float* g_xyz, float* g_centres, float* g_ibeta, void* stream);

/* Per-frame tall-skinny product out[m][i][j] += sum_{s in frame m} A[s][i] * Bm[s][j]  (A: (S,CA<=80), Bm: (S,CB<=16), CA*CB<=640,
 * out: (M,CA,CB), accumulated).  All per-frame parameter gradients of the skinning warp reduce to this form
 * (bone transforms are affine in the point, the blended dual quaternion is linear in the skin weights). */
int lab4d_gram_per_frame(const float* A, int CA, const float* Bm, int CB, int S, int spf, int M, float* out, void* stream);

/* The gaussian-scaled bone coordinates of a point are affine in the point: c[b][k] = aff[m][b][k][0..2] . x + aff[m][b][k][3] with
 * aff = (M, B, 3, 4) fp32 built from the bone-to-object dual quaternions and the gaussian scales (transforms.py:9-25,
 * skinning.py:126-140).  The blend kernels build the table internally; this entry hands it to the delta-skin chain
 * (lab4d_mlp_fwd_args.aff), which then forms its 3B inputs in the kernel instead of reading an (S,3B) tensor. */
int lab4d_bone_affine(const float* art_r, const float* art_d, const float* gauss, int M, int B, float* aff, void* stream);

#endif /* LAB4D_SKIN_H */
