// Skeleton forward kinematics, one row (= one frame) at a time: joint angles -> bone dual quaternions, and the adjoint.
//
// Replaces (paths relative to lab4d/):
//   utils/geom_utils.py:110-140      so3_to_exp_map          (Rodrigues, theta clamped from below)
//   utils/skel_utils.py:50-103       fk_se3                  (walk `edges` in dict order: global = parent_global @ local)
//   utils/quat_transform.py:468-532  matrix_to_quaternion    (best-conditioned of four candidates)
//   utils/skel_utils.py:106-145      shift_joints_to_bones_dq / shift_joints_to_bones
//   nnutils/pose.py:472-502          ArticulationSkelMLP.compute_rel_rest_joints (symmetrised bone lengths)
// The reference runs this as a 25-iteration Python loop over (..,4,4) matrices with clone / index_put per joint (~150
// launches forward, more in backward); here one thread owns one row and keeps the whole tree in its private arrays.
//
// The functions are plain C++ (LAB4D_HD = __host__ __device__ under hipcc, nothing otherwise) so that the CPU test-suite
// can compile this very header with g++ and hold the arithmetic to the oracle without a GPU (tests/host_harness/): that
// harness is test infrastructure, the product only ever launches the kernels of fk.hip.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define LAB4D_HD __host__ __device__ inline
#else
#define LAB4D_HD inline
#endif

namespace lab4d_fk {

constexpr int MAXB = 32;

struct Skel {
    int B;
    const int* order;   // joints in the reference's visiting order (keys of `edges`, 0-based); B entries
    const int* parent;  // parent[j]: 0-based parent joint of j, -1 when the parent is the root
    const int* symm;    // symm[j]: symmetric partner of j (bone lengths only; may be null when lengths are not used)
};

// ---- 3x3 helpers (row-major) ----------------------------------------------------------------------------------------
LAB4D_HD void mat_mul(const float* A, const float* Bm, float* C) {  // C = A B
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * Bm[j] + A[3 * i + 1] * Bm[3 + j] + A[3 * i + 2] * Bm[6 + j];
}
LAB4D_HD void mat_tmul(const float* A, const float* Bm, float* C) {  // C = A^T B
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * Bm[j] + A[3 + i] * Bm[3 + j] + A[6 + i] * Bm[6 + j];
}
LAB4D_HD void mat_vec(const float* A, const float* v, float* o) {
    for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
LAB4D_HD void mat_tvec(const float* A, const float* v, float* o) {
    for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}

// ---- so3_to_exp_map: R = I + sin(th) V + (1 - cos(th)) V^2, v = w / th, th = max(|w|, 1e-6), V = hat(v) -------------
LAB4D_HD void exp_map_fwd(const float* w, float* R) {
    const float n = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const float th = fmaxf(n, 1e-6f), inv = 1.f / th;
    const float v0 = w[0] * inv, v1 = w[1] * inv, v2 = w[2] * inv;
    const float s = sinf(th), c1 = 1.f - cosf(th), vv = v0 * v0 + v1 * v1 + v2 * v2;
    R[0] = 1.f + c1 * (v0 * v0 - vv); R[1] = -s * v2 + c1 * v0 * v1;     R[2] = s * v1 + c1 * v0 * v2;
    R[3] = s * v2 + c1 * v1 * v0;     R[4] = 1.f + c1 * (v1 * v1 - vv);  R[5] = -s * v0 + c1 * v1 * v2;
    R[6] = -s * v1 + c1 * v2 * v0;    R[7] = s * v0 + c1 * v2 * v1;      R[8] = 1.f + c1 * (v2 * v2 - vv);
}

LAB4D_HD void exp_map_bwd(const float* w, const float* g, float* gw) {
    const float n = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const float th = fmaxf(n, 1e-6f), inv = 1.f / th;
    const float v[3] = {w[0] * inv, w[1] * inv, w[2] * inv};
    const float s = sinf(th), c = cosf(th), c1 = 1.f - c, vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const float tr = g[0] + g[4] + g[8];
    // d/ds : <g, V>;  d/dc1 : <g, v v^T - vv I>
    const float gs = v[0] * (g[7] - g[5]) + v[1] * (g[2] - g[6]) + v[2] * (g[3] - g[1]);
    float gc1 = -vv * tr, gv[3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) gc1 += g[3 * a + b] * v[a] * v[b];
    gv[0] = s * (g[7] - g[5]);
    gv[1] = s * (g[2] - g[6]);
    gv[2] = s * (g[3] - g[1]);
    for (int a = 0; a < 3; ++a) {
        float acc = 0.f;
        for (int b = 0; b < 3; ++b) acc += (g[3 * a + b] + g[3 * b + a]) * v[b];
        gv[a] += c1 * (acc - 2.f * v[a] * tr);
    }
    float gth = gs * c + gc1 * s;  // d sin = cos, d(1 - cos) = sin
    gth -= (gv[0] * w[0] + gv[1] * w[1] + gv[2] * w[2]) * inv * inv;
    const float k = (n >= 1e-6f) ? gth / n : 0.f;  // clamp passes the gradient only above the floor; d|w|/dw = w/|w|
    for (int a = 0; a < 3; ++a) gw[a] = gv[a] * inv + k * w[a];
}

// ---- matrix_to_quaternion ---------------------------------------------------------------------------------------------
// candidate i = quaternion * (its i-th component); numerators are linear in m except the i-th one (= q_abs_i^2).
LAB4D_HD int quat_branch(const float* m, float* s4) {
    s4[0] = 1.f + m[0] + m[4] + m[8];
    s4[1] = 1.f + m[0] - m[4] - m[8];
    s4[2] = 1.f - m[0] + m[4] - m[8];
    s4[3] = 1.f - m[0] - m[4] + m[8];
    int best = 0;
    float bv = s4[0] > 0.f ? s4[0] : 0.f;
    for (int i = 1; i < 4; ++i) {
        const float x = s4[i] > 0.f ? s4[i] : 0.f;
        if (x > bv) { bv = x; best = i; }  // first maximum wins, like torch.argmax (sqrt is monotone)
    }
    return best;
}
LAB4D_HD void quat_numerators(const float* m, int i, float* n) {
    if (i == 0)      { n[0] = 0.f; n[1] = m[7] - m[5]; n[2] = m[2] - m[6]; n[3] = m[3] - m[1]; }
    else if (i == 1) { n[0] = m[7] - m[5]; n[1] = 0.f; n[2] = m[3] + m[1]; n[3] = m[2] + m[6]; }
    else if (i == 2) { n[0] = m[2] - m[6]; n[1] = m[3] + m[1]; n[2] = 0.f; n[3] = m[5] + m[7]; }
    else             { n[0] = m[3] - m[1]; n[1] = m[6] + m[2]; n[2] = m[7] + m[5]; n[3] = 0.f; }
}
LAB4D_HD void mat_to_quat_fwd(const float* m, float* q) {
    float s4[4], n[4];
    const int i = quat_branch(m, s4);
    const float a = s4[i] > 0.f ? sqrtf(s4[i]) : 0.f;
    quat_numerators(m, i, n);
    n[i] = a * a;
    const float inv = 1.f / (2.f * fmaxf(a, 0.1f));
    for (int k = 0; k < 4; ++k) q[k] = n[k] * inv;
}
LAB4D_HD void mat_to_quat_bwd(const float* m, const float* gq, float* gm) {
    float s4[4], n[4];
    const int i = quat_branch(m, s4);
    for (int k = 0; k < 9; ++k) gm[k] = 0.f;
    if (!(s4[i] > 0.f)) return;
    const float a = sqrtf(s4[i]);
    quat_numerators(m, i, n);
    float gn[4], ga;
    if (a >= 0.1f) {  // q_i = a/2, q_k = n_k / (2a)
        const float inv = 0.5f / a;
        ga = 0.5f * gq[i];
        for (int k = 0; k < 4; ++k) {
            gn[k] = gq[k] * inv;
            if (k != i) ga -= gq[k] * n[k] * inv / a;
        }
    } else {  // floored denominator (never the case for a rotation matrix: max_i s_i >= 1)
        for (int k = 0; k < 4; ++k) gn[k] = gq[k] * 5.f;
        ga = 2.f * a * gn[i];
    }
    gn[i] = 0.f;
    const float gsv = ga * 0.5f / a;  // s_i = 1 +- m00 +- m11 +- m22
    const float sg[4][3] = {{1.f, 1.f, 1.f}, {1.f, -1.f, -1.f}, {-1.f, 1.f, -1.f}, {-1.f, -1.f, 1.f}};
    gm[0] = sg[i][0] * gsv; gm[4] = sg[i][1] * gsv; gm[8] = sg[i][2] * gsv;
    // off-diagonal numerators, see quat_numerators
    if (i == 0)      { gm[7] += gn[1]; gm[5] -= gn[1]; gm[2] += gn[2]; gm[6] -= gn[2]; gm[3] += gn[3]; gm[1] -= gn[3]; }
    else if (i == 1) { gm[7] += gn[0]; gm[5] -= gn[0]; gm[3] += gn[2]; gm[1] += gn[2]; gm[2] += gn[3]; gm[6] += gn[3]; }
    else if (i == 2) { gm[2] += gn[0]; gm[6] -= gn[0]; gm[3] += gn[1]; gm[1] += gn[1]; gm[5] += gn[3]; gm[7] += gn[3]; }
    else             { gm[3] += gn[0]; gm[1] -= gn[0]; gm[6] += gn[1]; gm[2] += gn[1]; gm[7] += gn[2]; gm[5] += gn[2]; }
}

// ---- bone lengths: loc_j = rest_local_j * (exp(l_j + ls) + exp(l_symm(j) + ls)) / 2 ---------------------------------
LAB4D_HD void local_joints_fwd(const Skel& sk, const float* rest_local, const float* loglen, float logscale, float* loc) {
    for (int j = 0; j < sk.B; ++j) {
        const float len = 0.5f * (expf(loglen[j] + logscale) + expf(loglen[sk.symm[j]] + logscale));
        for (int a = 0; a < 3; ++a) loc[3 * j + a] = rest_local[3 * j + a] * len;
    }
}
LAB4D_HD void local_joints_bwd(const Skel& sk, const float* rest_local, const float* loglen, float logscale, const float* gloc,
                               float* g_loglen, float* g_logscale) {
    float gls = 0.f;
    for (int j = 0; j < sk.B; ++j) g_loglen[j] = 0.f;
    for (int j = 0; j < sk.B; ++j) {
        const float glen = gloc[3 * j] * rest_local[3 * j] + gloc[3 * j + 1] * rest_local[3 * j + 1] + gloc[3 * j + 2] * rest_local[3 * j + 2];
        const float e0 = 0.5f * glen * expf(loglen[j] + logscale), e1 = 0.5f * glen * expf(loglen[sk.symm[j]] + logscale);
        g_loglen[j] += e0;
        g_loglen[sk.symm[j]] += e1;
        gls += e0 + e1;
    }
    *g_logscale = gls;
}

// ---- the kinematic chain --------------------------------------------------------------------------------------------
LAB4D_HD void chain_fwd(const Skel& sk, const float* so3, const float* loc, float* GR, float* Gt) {
    for (int j = 0; j < sk.B; ++j) {
        for (int k = 0; k < 9; ++k) GR[9 * j + k] = (k % 4 == 0) ? 1.f : 0.f;
        Gt[3 * j] = Gt[3 * j + 1] = Gt[3 * j + 2] = 0.f;
    }
    for (int p = 0; p < sk.B; ++p) {
        const int j = sk.order[p], par = sk.parent[j];
        float R[9];
        exp_map_fwd(so3 + 3 * j, R);
        if (par >= 0) {  // the parent's *current* global transform (identity if it has not been visited yet)
            float PR[9], Pt[3], t[3];
            for (int k = 0; k < 9; ++k) PR[k] = GR[9 * par + k];
            for (int k = 0; k < 3; ++k) Pt[k] = Gt[3 * par + k];
            mat_mul(PR, R, GR + 9 * j);
            mat_vec(PR, loc + 3 * j, t);
            for (int k = 0; k < 3; ++k) Gt[3 * j + k] = t[k] + Pt[k];
        } else {
            for (int k = 0; k < 9; ++k) GR[9 * j + k] = R[k];
            for (int k = 0; k < 3; ++k) Gt[3 * j + k] = loc[3 * j + k];
        }
    }
}

// gGR / gGt: gradients wrt the global transforms (consumed: the parents' entries are accumulated into)
LAB4D_HD void chain_bwd(const Skel& sk, const float* so3, const float* loc, const float* GR, float* gGR, float* gGt, float* g_so3,
                        float* g_loc) {
    int pos[MAXB];
    for (int j = 0; j < sk.B; ++j) {
        pos[j] = MAXB;
        g_so3[3 * j] = g_so3[3 * j + 1] = g_so3[3 * j + 2] = 0.f;
        g_loc[3 * j] = g_loc[3 * j + 1] = g_loc[3 * j + 2] = 0.f;
    }
    for (int p = 0; p < sk.B; ++p) pos[sk.order[p]] = p;
    for (int p = sk.B - 1; p >= 0; --p) {
        const int j = sk.order[p], par = sk.parent[j];
        float R[9], gR[9];
        exp_map_fwd(so3 + 3 * j, R);
        if (par >= 0 && pos[par] < p) {
            const float* PR = GR + 9 * par;
            mat_tmul(PR, gGR + 9 * j, gR);
            mat_tvec(PR, gGt + 3 * j, g_loc + 3 * j);
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) {  // gP_R += gG R^T + gGt loc^T
                    float acc = gGt[3 * j + a] * loc[3 * j + b];
                    for (int c = 0; c < 3; ++c) acc += gGR[9 * j + 3 * a + c] * R[3 * b + c];
                    gGR[9 * par + 3 * a + b] += acc;
                }
                gGt[3 * par + a] += gGt[3 * j + a];
            }
        } else {  // parent = root, or not yet visited at that point: identity
            for (int k = 0; k < 9; ++k) gR[k] = gGR[9 * j + k];
            for (int k = 0; k < 3; ++k) g_loc[3 * j + k] = gGt[3 * j + k];
        }
        exp_map_bwd(so3 + 3 * j, gR, g_so3 + 3 * j);
    }
}

// ---- joints -> (bone) dual quaternions ------------------------------------------------------------------------------
// centre_p = mean over children c of (J_p + J_c)/2 for joints with children, J_p for leaves (J = joint position + shift)
LAB4D_HD void bone_centres_fwd(const Skel& sk, const float* Gt, const float* shift, int bones, float* ctr) {
    float acc[MAXB * 3];
    int cnt[MAXB];
    for (int j = 0; j < sk.B; ++j) { cnt[j] = 0; acc[3 * j] = acc[3 * j + 1] = acc[3 * j + 2] = 0.f; }
    const float s0 = shift ? shift[0] : 0.f, s1 = shift ? shift[1] : 0.f, s2 = shift ? shift[2] : 0.f;
    if (bones)
        for (int j = 0; j < sk.B; ++j) {
            const int par = sk.parent[j];
            if (par < 0) continue;
            cnt[par]++;
            acc[3 * par] += Gt[3 * j] + s0; acc[3 * par + 1] += Gt[3 * j + 1] + s1; acc[3 * par + 2] += Gt[3 * j + 2] + s2;
        }
    for (int j = 0; j < sk.B; ++j) {
        const float J[3] = {Gt[3 * j] + s0, Gt[3 * j + 1] + s1, Gt[3 * j + 2] + s2};
        const float k = cnt[j] ? 0.5f / (float)cnt[j] : 0.f, h = cnt[j] ? 0.5f : 1.f;
        for (int a = 0; a < 3; ++a) ctr[3 * j + a] = h * J[a] + k * acc[3 * j + a];
    }
}
LAB4D_HD void bone_centres_bwd(const Skel& sk, const float* gctr, int bones, float* gGt, float* g_shift) {
    int cnt[MAXB];
    for (int j = 0; j < sk.B; ++j) cnt[j] = 0;
    if (bones)
        for (int j = 0; j < sk.B; ++j)
            if (sk.parent[j] >= 0) cnt[sk.parent[j]]++;
    for (int j = 0; j < sk.B; ++j) {
        const float h = cnt[j] ? 0.5f : 1.f;
        for (int a = 0; a < 3; ++a) gGt[3 * j + a] = h * gctr[3 * j + a];
    }
    if (bones)
        for (int j = 0; j < sk.B; ++j) {
            const int par = sk.parent[j];
            if (par < 0) continue;
            const float k = 0.5f / (float)cnt[par];
            for (int a = 0; a < 3; ++a) gGt[3 * j + a] += k * gctr[3 * par + a];
        }
    g_shift[0] = g_shift[1] = g_shift[2] = 0.f;
    for (int j = 0; j < sk.B; ++j)
        for (int a = 0; a < 3; ++a) g_shift[a] += gGt[3 * j + a];
}

// qd = 0.5 * (0, c) * q   (quaternion_translation_to_dual_quaternion, quat_transform.py:290-297)
LAB4D_HD void dual_part_fwd(const float* c, const float* q, float* qd) {
    qd[0] = 0.5f * (-c[0] * q[1] - c[1] * q[2] - c[2] * q[3]);
    qd[1] = 0.5f * (c[0] * q[0] + c[1] * q[3] - c[2] * q[2]);
    qd[2] = 0.5f * (-c[0] * q[3] + c[1] * q[0] + c[2] * q[1]);
    qd[3] = 0.5f * (c[0] * q[2] - c[1] * q[1] + c[2] * q[0]);
}
LAB4D_HD void dual_part_bwd(const float* c, const float* q, const float* g, float* gc, float* gq /* += */) {
    gc[0] = 0.5f * (-g[0] * q[1] + g[1] * q[0] - g[2] * q[3] + g[3] * q[2]);
    gc[1] = 0.5f * (-g[0] * q[2] + g[1] * q[3] + g[2] * q[0] - g[3] * q[1]);
    gc[2] = 0.5f * (-g[0] * q[3] - g[1] * q[2] + g[2] * q[1] + g[3] * q[0]);
    gq[0] += 0.5f * (g[1] * c[0] + g[2] * c[1] + g[3] * c[2]);
    gq[1] += 0.5f * (-g[0] * c[0] + g[2] * c[2] - g[3] * c[1]);
    gq[2] += 0.5f * (-g[0] * c[1] - g[1] * c[2] + g[3] * c[0]);
    gq[3] += 0.5f * (-g[0] * c[2] + g[1] * c[1] - g[2] * c[0]);
}

// ---- whole row ----------------------------------------------------------------------------------------------------------
// so3 (B,3), loc (B,3) local joints, shift (3) or null; bones != 0: shift_joints_to_bones_dq on top of fk_se3.
LAB4D_HD void row_forward(const Skel& sk, const float* so3, const float* loc, const float* shift, int bones, float* qr, float* qd) {
    float GR[MAXB * 9], Gt[MAXB * 3], ctr[MAXB * 3];
    chain_fwd(sk, so3, loc, GR, Gt);
    bone_centres_fwd(sk, Gt, shift, bones, ctr);
    for (int j = 0; j < sk.B; ++j) {
        mat_to_quat_fwd(GR + 9 * j, qr + 4 * j);
        dual_part_fwd(ctr + 3 * j, qr + 4 * j, qd + 4 * j);
    }
}

LAB4D_HD void row_backward(const Skel& sk, const float* so3, const float* loc, const float* shift, int bones, const float* g_qr,
                           const float* g_qd, float* g_so3, float* g_loc, float* g_shift) {
    float GR[MAXB * 9], Gt[MAXB * 3], ctr[MAXB * 3], gGR[MAXB * 9], gGt[MAXB * 3], gctr[MAXB * 3];
    chain_fwd(sk, so3, loc, GR, Gt);
    bone_centres_fwd(sk, Gt, shift, bones, ctr);
    for (int j = 0; j < sk.B; ++j) {
        float q[4], gq[4] = {g_qr[4 * j], g_qr[4 * j + 1], g_qr[4 * j + 2], g_qr[4 * j + 3]};
        mat_to_quat_fwd(GR + 9 * j, q);
        dual_part_bwd(ctr + 3 * j, q, g_qd + 4 * j, gctr + 3 * j, gq);
        mat_to_quat_bwd(GR + 9 * j, gq, gGR + 9 * j);
    }
    bone_centres_bwd(sk, gctr, bones, gGt, g_shift);
    chain_bwd(sk, so3, loc, GR, gGR, gGt, g_so3, g_loc);
}

}  // namespace lab4d_fk
