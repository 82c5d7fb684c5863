/*
 * lab4d_pose.h -- skeleton forward kinematics of the per-frame articulation path (included by lab4d_hip.h).
 * the first "next" row after the per-sample hot path.
 *
 * Replaces (paths relative to lab4d/):
 *   utils/geom_utils.py:110-140      so3_to_exp_map
 *   utils/skel_utils.py:50-103       fk_se3(local_rest_joints, so3, edges, to_dq=True)
 *   utils/quat_transform.py:468-532  matrix_to_quaternion
 *   utils/skel_utils.py:106-145      shift_joints_to_bones_dq(dq, edges, shift)
 *   nnutils/pose.py:472-502          ArticulationSkelMLP.compute_rel_rest_joints
 * as called from ArticulationSkelMLP.forward / get_vals_and_mean (nnutils/pose.py:417-470,526-573).  The reference walks the
 * kinematic tree in a Python loop over (..,4,4) matrices (one clone + matmul + index_put per joint, then ~25 kernels of
 * matrix_to_quaternion / dual-quaternion conversion); here a row (= one frame) is one thread and the call is one launch.
 *
 * The skeleton is data: `order` = the keys of the reference's `edges` dict in iteration order, 0-based (B entries);
 * `parent[j]` = 0-based parent of joint j, -1 when the parent is the root; `symm[j]` = symmetric partner of j.
 * B <= 32.  All fp32; R rows.  Outputs are dual quaternions ((R,B,4) real, (R,B,4) dual), like the reference.
 */
#ifndef LAB4D_POSE_H
#define LAB4D_POSE_H

/* fk_se3 (bones == 0, shift == NULL) or fk_se3 + shift_joints_to_bones_dq (bones != 0, shift (3) or NULL).
 * so3, local: (R,B,3). */
int lab4d_fk_forward(const float* so3, const float* local, const float* shift, const int32_t* order, const int32_t* parent, int R,
                     int B, int bones, float* qr, float* qd, void* stream);
/* Adjoint: g_qr, g_qd (R,B,4) -> g_so3, g_local (R,B,3) written; g_shift (R,3) per-row partials written (may be NULL; the
 * caller sums over rows -- deterministic, no atomics). */
int lab4d_fk_backward(const float* so3, const float* local, const float* shift, const int32_t* order, const int32_t* parent,
                      const float* g_qr, const float* g_qd, int R, int B, int bones, float* g_so3, float* g_local, float* g_shift,
                      void* stream);

/* ArticulationSkelMLP.forward after the so3 head, fused: local joints = rest_local (B,3) * (exp(loglen_j + logscale) +
 * exp(loglen_symm(j) + logscale)) / 2 with loglen (R,B) the per-row log-bone-length increments and logscale a device scalar;
 * then fk_se3 and shift_joints_to_bones_dq(shift (3)). */
int lab4d_skel_bones_forward(const float* so3, const float* loglen, const float* logscale, const float* rest_local,
                             const float* shift, const int32_t* order, const int32_t* parent, const int32_t* symm, int R, int B,
                             float* qr, float* qd, void* stream);
/* Adjoint: g_so3 (R,B,3), g_loglen (R,B), g_logscale (R) and g_shift (R,3; may be NULL) per-row partials, all written. */
int lab4d_skel_bones_backward(const float* so3, const float* loglen, const float* logscale, const float* rest_local,
                              const float* shift, const int32_t* order, const int32_t* parent, const int32_t* symm,
                              const float* g_qr, const float* g_qd, int R, int B, float* g_so3, float* g_loglen,
                              float* g_logscale, float* g_shift, void* stream);

#endif /* LAB4D_POSE_H */
