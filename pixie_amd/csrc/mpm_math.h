// pixie_amd/csrc/mpm_math.h -- per-particle MLS-MPM arithmetic for the gfx950 kernels.
//
// Everything here is register-level fp32 math (no memory access) written for the fused
// HIP particle kernel in mpm_kernels.hip.  It is also compilable as plain C++ so that
// tests/host_harness can check each function against oracle/ on the CPU; the product
// never runs it on the host.
//
// What is computed follows the reference's Warp device functions
// (third_party/PhysGaussian/mpm_solver_warp/mpm_utils.py, cited per function); how it is
// computed is ours: the 3x3 SVD is a fixed-sweep Jacobi on F^T F with a Gram-Schmidt/cross
// product U, kept branch-light so that a 64-wide wavefront does not diverge.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define PX_HD __host__ __device__ __forceinline__
#define PX_RARE static __host__ __device__ __forceinline__   // rare paths: out of line, so that their registers stay out of the hot kernels' budget
#else
#define PX_HD inline
#define PX_RARE inline
#endif

#ifndef PX_SVD_NEWTON_SCHULZ
#define PX_SVD_NEWTON_SCHULZ 0
#endif

namespace pixie {

struct Mat3 {
    float m[9];  // row-major, m[3*r+c]  (warp mat33 layout, warp_utils.py:48)
    PX_HD float& operator()(int r, int c) { return m[3 * r + c]; }
    PX_HD float operator()(int r, int c) const { return m[3 * r + c]; }
};
struct Vec3 {
    float v[3];
};

PX_HD Mat3 mat_identity() {
    Mat3 r;
    for (int i = 0; i < 9; ++i) r.m[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    return r;
}
PX_HD Mat3 mat_mul(const Mat3& a, const Mat3& b) {
    Mat3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return r;
}
// a * b^T
PX_HD Mat3 mat_mul_bt(const Mat3& a, const Mat3& b) {
    Mat3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.m[3 * i + j] = a.m[3 * i] * b.m[3 * j] + a.m[3 * i + 1] * b.m[3 * j + 1] + a.m[3 * i + 2] * b.m[3 * j + 2];
    return r;
}
PX_HD float mat_det(const Mat3& a) {
    return a.m[0] * (a.m[4] * a.m[8] - a.m[5] * a.m[7]) - a.m[1] * (a.m[3] * a.m[8] - a.m[5] * a.m[6]) +
           a.m[2] * (a.m[3] * a.m[7] - a.m[4] * a.m[6]);
}
// U * diag(d) * V^T
PX_HD Mat3 mat_udvt(const Mat3& U, const float d[3], const Mat3& V) {
    Mat3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.m[3 * i + j] = U.m[3 * i] * d[0] * V.m[3 * j] + U.m[3 * i + 1] * d[1] * V.m[3 * j + 1] +
                             U.m[3 * i + 2] * d[2] * V.m[3 * j + 2];
    return r;
}

// 1-ulp reciprocal / square root (v_rcp_f32, v_sqrt_f32) for the self-correcting polar iteration; IEEE on the host build
PX_HD float px_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
PX_HD float px_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
PX_HD float px_rsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return rsqrtf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}
// The bare hardware instructions for arguments known to be normal numbers (1 ulp each; the library forms wrap them in
// denormal scaling and extended-precision range reduction, ~10 VALU instructions apiece in a VALU-issue-bound kernel):
// v_rsq_f32, a * v_rcp_f32(b), v_log_f32 * ln 2, v_exp_f32(x * log2 e).  IEEE on the host build.
PX_HD float px_rsq_normal(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}
PX_HD float px_div(float a, float b) { return a * px_rcp(b); }
PX_HD float px_log(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __logf(x);
#else
    return logf(x);
#endif
}
PX_HD float px_exp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __expf(x);
#else
    return expf(x);
#endif
}

// One Jacobi rotation on the symmetric matrix S (full Mat3 storage) annihilating S(P,Q), accumulated into V.
// The angle comes from v_sqrt_f32 / v_rcp_f32 (1 ulp): an inexact angle only slows the convergence of a self-correcting
// iteration, while c^2 + s^2 = 1 -- what the orthogonality of V rests on -- holds to the accuracy of the one v_rsq_f32.
// (IEEE sqrtf and '/' cost ~10 VALU instructions each on gfx950; this kernel is VALU-issue bound.)
// `frozen` lanes (converged earlier) pass through bit-for-bit: t = 0 gives c = 1, s = 0 and every update is x*1 - y*0.
template <int P, int Q>
PX_HD void jacobi_rotate(Mat3& S, Mat3& V, bool frozen = false) {
    const float apq = S(P, Q);
    const float app = S(P, P), aqq = S(Q, Q);
    // theta = (aqq-app)/(2apq); t = sgn(theta)/(|theta|+sqrt(theta^2+1)) = 2apq / (d + sgn(d) sqrt(d^2 + (2apq)^2)), |t| <= 1
    const float d = aqq - app;
    const float two_apq = 2.0f * apq;
    const float h = px_sqrt(d * d + two_apq * two_apq);
    const float denom = d + copysignf(h, d);
    float t = two_apq * px_rcp(denom);
    // negligible off-diagonal (also: 0/0 when d = apq = 0, and NaN input): no rotation
    if (frozen || !(fabsf(apq) > 1e-12f * (fabsf(app) + fabsf(aqq)))) t = 0.0f;
    const float c = (t == 0.0f) ? 1.0f : px_rsq_normal(t * t + 1.0f);   // argument in [1, 2]
    const float s = t * c;
    // S <- J^T S J with J = [[c, s],[-s, c]] on (P,Q)
    constexpr int R = 3 - P - Q;
    const float arp = S(R, P), arq = S(R, Q);
    S(P, P) = app - t * apq;
    S(Q, Q) = aqq + t * apq;
    if (!frozen) { S(P, Q) = 0.0f; S(Q, P) = 0.0f; }
    const float nrp = c * arp - s * arq;
    const float nrq = s * arp + c * arq;
    S(R, P) = nrp; S(P, R) = nrp;
    S(R, Q) = nrq; S(Q, R) = nrq;
    for (int i = 0; i < 3; ++i) {
        const float vp = V(i, P), vq = V(i, Q);
        V(i, P) = c * vp - s * vq;
        V(i, Q) = s * vp + c * vq;
    }
}

// swap columns a,b of B and V, negating one so det(V) is preserved
PX_HD void cond_swap_cols(bool doit, Mat3& B, Mat3& V, float& na, float& nb, int a, int b) {
    if (doit) {
        for (int i = 0; i < 3; ++i) {
            float t = B(i, a); B(i, a) = B(i, b); B(i, b) = -t;
            t = V(i, a); V(i, a) = V(i, b); V(i, b) = -t;
        }
        float t = na; na = nb; nb = t;
    }
}

// 3x3 SVD in the convention the reference relies on from wp.svd3 (Warp built-in, called at
// mpm_utils.py:94,145,202,249,501,566): F = U diag(sig) V^T with U, V proper rotations,
// |sig0| >= |sig1| >= |sig2| and only sig2 allowed to be negative (sign(sig2) = sign(det F)).
//
// Cyclic Jacobi on F^T F, sweeps until THIS lane's off-diagonal mass is below float32 resolution
// (off^2 <= 1e-14 tr^2: a reconstruction error of 1e-7 |F|, singular values second order in it), at most kSvdSweeps.
// A lane freezes at its own last sweep, so its result does not depend on which other particles share its wave; the wave
// leaves the loop when every lane has frozen -- 2 sweeps for the strains of a stable simulation, 3 for anything else
// that is not pathological (tests/test_mpm_oracle.py prints the histogram).
constexpr int kSvdSweeps = 6;
PX_HD bool svd_converged(const Mat3& S) {
    const float off2 = S(0, 1) * S(0, 1) + S(0, 2) * S(0, 2) + S(1, 2) * S(1, 2);
    const float tr = S(0, 0) + S(1, 1) + S(2, 2);
    return !(off2 > 1.0e-14f * tr * tr);   // NaN counts as converged (nothing sensible can be done with it)
}
PX_HD void svd3(const Mat3& F, Mat3& U, float sig[3], Mat3& V, int* sweeps_out = nullptr) {
    Mat3 S;  // F^T F
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) S(i, j) = F(0, i) * F(0, j) + F(1, i) * F(1, j) + F(2, i) * F(2, j);
    V = mat_identity();
    bool frozen = svd_converged(S);
    int sweeps = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int sweep = 0; sweep < kSvdSweeps; ++sweep) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (__all(frozen)) break;
#else
        if (frozen) break;
#endif
        jacobi_rotate<0, 1>(S, V, frozen);
        jacobi_rotate<0, 2>(S, V, frozen);
        jacobi_rotate<1, 2>(S, V, frozen);
        if (!frozen) ++sweeps;
        frozen = frozen || svd_converged(S);
    }
    if (sweeps_out) *sweeps_out = sweeps;
#if PX_SVD_NEWTON_SCHULZ
    // accumulated fp32 rotations leave V orthogonal only to a few 1e-7; one Newton-Schulz step
    // V <- V (3I - V^T V)/2 squares that error.
    {
        Mat3 G;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const float g = V(0, i) * V(0, j) + V(1, i) * V(1, j) + V(2, i) * V(2, j);
                G(i, j) = ((i == j) ? 1.5f : 0.0f) - 0.5f * g;
            }
        V = mat_mul(V, G);
    }
#endif
    Mat3 B = mat_mul(F, V);
    float n0 = B(0, 0) * B(0, 0) + B(1, 0) * B(1, 0) + B(2, 0) * B(2, 0);
    float n1 = B(0, 1) * B(0, 1) + B(1, 1) * B(1, 1) + B(2, 1) * B(2, 1);
    float n2 = B(0, 2) * B(0, 2) + B(1, 2) * B(1, 2) + B(2, 2) * B(2, 2);
    cond_swap_cols(n0 < n1, B, V, n0, n1, 0, 1);
    cond_swap_cols(n0 < n2, B, V, n0, n2, 0, 2);
    cond_swap_cols(n1 < n2, B, V, n1, n2, 1, 2);
    // U by Gram-Schmidt on the (already nearly orthogonal) columns of B; u2 = u0 x u1
    float u0[3], u1[3], u2[3];
    const float i0 = (n0 > 1e-36f) ? px_rsqrt(n0) : 0.0f;
    for (int i = 0; i < 3; ++i) u0[i] = B(i, 0) * i0;
    if (n0 <= 1e-36f) { u0[0] = 1.0f; u0[1] = 0.0f; u0[2] = 0.0f; }
    const float d01 = u0[0] * B(0, 1) + u0[1] * B(1, 1) + u0[2] * B(2, 1);
    for (int i = 0; i < 3; ++i) u1[i] = B(i, 1) - d01 * u0[i];
    float m1 = u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2];
    if (m1 <= 1e-36f) {  // rank <= 1: any unit vector orthogonal to u0
        const float ax = fabsf(u0[0]), ay = fabsf(u0[1]), az = fabsf(u0[2]);
        float e[3] = {0.0f, 0.0f, 0.0f};
        e[(ax <= ay && ax <= az) ? 0 : (ay <= az ? 1 : 2)] = 1.0f;
        const float de = e[0] * u0[0] + e[1] * u0[1] + e[2] * u0[2];
        for (int i = 0; i < 3; ++i) u1[i] = e[i] - de * u0[i];
        m1 = u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2];
    }
    const float i1 = px_rsqrt(m1);
    for (int i = 0; i < 3; ++i) u1[i] *= i1;
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
    u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
    u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
    for (int i = 0; i < 3; ++i) { U(i, 0) = u0[i]; U(i, 1) = u1[i]; U(i, 2) = u2[i]; }
    sig[0] = u0[0] * B(0, 0) + u0[1] * B(1, 0) + u0[2] * B(2, 0);
    sig[1] = u1[0] * B(0, 1) + u1[1] * B(1, 1) + u1[2] * B(2, 1);
    sig[2] = u2[0] * B(0, 2) + u2[1] * B(1, 2) + u2[2] * B(2, 2);
}

// Crushed elements (one stretch below half the largest): forming F F^T squares the condition number, so the Jacobi iteration on it
// resolves the eigen-frame only to ~1e-7 lam_max ABSOLUTE -- two small axes keep a mutual rotation of 1e-7 lam_max / (lam_p - lam_q)
// (sigma = (1, 1e-2, 2e-2): 3e-4 rad), and sqrt(lam_d) an error of 1e-7 lam_max / sig_d.  The columns a_d = F^T u_d
// (= sig_d v_d) come from F itself and are accurate relative to |F|: one sweep of one-sided (Hestenes) rotations that make them
// orthogonal, applied to U alike, finishes the frame; sig_d = |a_d|, a sum of squares, accurate relative to sig_d itself.
// Rare and out of line: keeps its registers out of the block kernel's budget.
struct FrameSig { Mat3 U; float sig[3]; };
// (Arguments and result BY VALUE and every index static after unrolling: an out-of-line function that takes references or indexes its
// arrays at run time puts its operands in scratch, and the scratch frame of the deepest call path is allocated for EVERY wave of the
// kernel -- 500 instead of 292 bytes per lane cost every MPM scene 20 %, jelly included; profiles/r6e_scratch_regression.txt.)
PX_RARE FrameSig refine_crushed_frame(Mat3 F, Mat3 U) {
    float a[3][3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int c = 0; c < 3; ++c) a[d][c] = F(0, c) * U(0, d) + F(1, c) * U(1, d) + F(2, c) * U(2, d);
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
        const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
        const float al = a[p][0] * a[p][0] + a[p][1] * a[p][1] + a[p][2] * a[p][2];
        const float be = a[q][0] * a[q][0] + a[q][1] * a[q][1] + a[q][2] * a[q][2];
        const float ga = a[p][0] * a[q][0] + a[p][1] * a[q][1] + a[p][2] * a[q][2];
        const bool rot = ga * ga > 1.0e-14f * al * be;              // else: orthogonal to float32 resolution already (or NaN / a null column)
        const float zeta = (be - al) / (2.0f * (rot ? ga : 1.0f));
        const float t = ((zeta >= 0.0f) ? 1.0f : -1.0f) / (fabsf(zeta) + px_sqrt(1.0f + zeta * zeta));
        const float c = rot ? px_rsqrt(1.0f + t * t) : 1.0f, sn = rot ? c * t : 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float ap = a[p][k], aq = a[q][k];
            a[p][k] = c * ap - sn * aq; a[q][k] = sn * ap + c * aq;
            const float up = U(k, p), uq = U(k, q);
            U(k, p) = c * up - sn * uq; U(k, q) = sn * up + c * uq;
        }
    }
    FrameSig r;
    r.U = U;
#pragma unroll
    for (int d = 0; d < 3; ++d) r.sig[d] = px_sqrt(a[d][0] * a[d][0] + a[d][1] * a[d][1] + a[d][2] * a[d][2]);
    return r;
}

// Left principal frame of F, the only part of the SVD the constitutive laws need:  F F^T = U diag(sig^2) U^T  by the same
// cyclic Jacobi iteration run on b = F F^T (accumulating U), sig_d = sqrt(lam_d), U a proper rotation, axes UNSORTED.
// For det F < 0 the sign goes to the singular value of smallest magnitude -- Warp's convention (see svd3), the one thing of
// the ordering that matters to callers: every law downstream is a function of the (sig_d, u_d) pairs and invariant under
// their permutation.  Against svd3 this drops the V accumulation's aftermath (B = F V, the column norms, the sort, the
// Gram-Schmidt U): ~170 of ~630 VALU instructions.  det_F = det F (the caller has it anyway).
PX_HD void left_stretch(const Mat3& F, float det_F, Mat3& U, float sig[3], int* sweeps_out = nullptr) {
    Mat3 S;  // F F^T
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) {
            const float v = F(i, 0) * F(j, 0) + F(i, 1) * F(j, 1) + F(i, 2) * F(j, 2);
            S(i, j) = v; S(j, i) = v;
        }
    U = mat_identity();
    bool frozen = svd_converged(S);
    int sweeps = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int sweep = 0; sweep < kSvdSweeps; ++sweep) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (__all(frozen)) break;
#else
        if (frozen) break;
#endif
        jacobi_rotate<0, 1>(S, U, frozen);
        jacobi_rotate<0, 2>(S, U, frozen);
        jacobi_rotate<1, 2>(S, U, frozen);
        if (!frozen) ++sweeps;
        frozen = frozen || svd_converged(S);
    }
    if (sweeps_out) *sweeps_out = sweeps;
    const float l0 = S(0, 0), l1 = S(1, 1), l2 = S(2, 2);
    sig[0] = px_sqrt(fmaxf(l0, 0.0f)); sig[1] = px_sqrt(fmaxf(l1, 0.0f)); sig[2] = px_sqrt(fmaxf(l2, 0.0f));
    // An eigenvalue of F F^T carries an absolute error of ~1e-7 lam_max, i.e. sig_d one of 1e-7 lam_max / sig_d: fine while the
    // stretches are within a factor 2 of each other (every stable simulation), not for a crushed element.  There (rare,
    // the lane's own test) the small axes are finished on F itself -- see refine_crushed_frame.
    if (fminf(fminf(l0, l1), l2) < 0.25f * fmaxf(fmaxf(l0, l1), l2)) {
        const FrameSig r = refine_crushed_frame(F, U);
        U = r.U;
        sig[0] = r.sig[0]; sig[1] = r.sig[1]; sig[2] = r.sig[2];
    }
    if (det_F < 0.0f) {
        if (sig[0] <= sig[1] && sig[0] <= sig[2]) sig[0] = -sig[0];
        else if (sig[1] <= sig[2]) sig[1] = -sig[1];
        else sig[2] = -sig[2];
    }
}

// Rotation factor R of the polar decomposition F = R S by the scaled Newton iteration
// R <- (g R + R^-T / g) / 2 (two Frobenius-scaled steps, then plain ones): a third of the instructions and of
// the dependency chain of svd3.  For det F > 0 it equals U V^T of the reference's wp.svd3 to fp32 roundoff, which is
// all kirchoff_stress_FCR (mpm_utils.py:10-17) needs.  Returns false when the iteration has not settled (extreme
// conditioning) or det F <= 0 (inverted element, where U V^T of the proper-rotation SVD is NOT the polar factor);
// the caller then takes the svd3 route.
//
// Stopping rule.  After ONE step every singular value of the iterate is >= 1 ((g s + 1/(g s))/2 >= 1), so from then on
// |R|_F^2 - 3 = sum (s_i^2 - 1) >= 2 max_i (s_i - 1): the Frobenius norm -- nine FMAs -- bounds the distance to the rotation.
// A step entered with max (s_i - 1) = e leaves e^2 / 2: an iterate with |R|_F^2 - 3 < 4e-4 (e < 2e-4) needs exactly one more
// step (2e-8, below fp32 roundoff), and that step is the lane's last.  A lane stops at ITS OWN last step (the iterate is
// frozen afterwards), so its result does not depend on which other particles share its wave; the wave leaves the loop once
// every lane has stopped -- after 2 of the 6 steps for strains below ~1.5 %, after 3 for the strains of a stable simulation.
PX_HD bool polar_rotation(const Mat3& F, Mat3& R) {
    R = F;
    float det = 1.0f;
    bool settled = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int it = 0; it < 6; ++it) {
        Mat3 cof;
        cof.m[0] = R.m[4] * R.m[8] - R.m[5] * R.m[7];
        cof.m[1] = R.m[5] * R.m[6] - R.m[3] * R.m[8];
        cof.m[2] = R.m[3] * R.m[7] - R.m[4] * R.m[6];
        cof.m[3] = R.m[2] * R.m[7] - R.m[1] * R.m[8];
        cof.m[4] = R.m[0] * R.m[8] - R.m[2] * R.m[6];
        cof.m[5] = R.m[1] * R.m[6] - R.m[0] * R.m[7];
        cof.m[6] = R.m[1] * R.m[5] - R.m[2] * R.m[4];
        cof.m[7] = R.m[2] * R.m[3] - R.m[0] * R.m[5];
        cof.m[8] = R.m[0] * R.m[4] - R.m[1] * R.m[3];
        const float d = R.m[0] * cof.m[0] + R.m[1] * cof.m[1] + R.m[2] * cof.m[2];
        if (it == 0) det = d;
        const float inv_d = px_rcp(d);
        float nr = 0.0f;
        for (int i = 0; i < 9; ++i) nr += R.m[i] * R.m[i];
        // the entering iterate is within 2e-4 of a rotation (valid from the second step on; NaN never is): last step
        const bool last = (it >= 1) && (nr - 3.0f < 4.0e-4f) && (nr - 3.0f > -1.0e-5f);
        float a = 0.5f, b = 0.5f * inv_d;  // R <- a R + b cof
        if (it < 2) {
            float nc = 0.0f;
            for (int i = 0; i < 9; ++i) nc += cof.m[i] * cof.m[i];
            // g^2 = |R^-T|_F / |R|_F = |cof|_F / (|d| |R|_F)   (only steers the convergence: approximate is fine)
            const float g2 = px_sqrt(nc * px_rcp(nr)) * fabsf(inv_d);
            const float g = px_sqrt(g2);
            a = 0.5f * g;
            b = 0.5f * inv_d * px_rcp(g);
        }
        for (int i = 0; i < 9; ++i) {
            const float r = a * R.m[i] + b * cof.m[i];
            if (!settled) R.m[i] = r;
        }
        if (last) settled = true;
#if defined(__HIP_DEVICE_COMPILE__)
        if (it >= 1 && __all(settled)) break;
#endif
    }
    return det > 0.0f && settled;
}

// ---------------------------------------------------------------- fixed-corotated stress without the rotation (small strain)
// kirchoff_stress_FCR (mpm_utils.py:10-17) is  tau = 2 mu (F - R) F^T + lam J (J - 1) I  with R the polar rotation of F.
// For det F > 0:  (F - R) F^T = F F^T - R F^T = b - sqrt(b),  b = F F^T  (R F^T = R S R^T is the left stretch, the symmetric
// positive square root of b).  With b = I + E:
//     b - sqrt(b) = E/2 + E^2/8 - E^3/16 + 5 E^4/128 - 7 E^5/256 + O(E^6)      (x - (sqrt(1 + x) - 1), coefficient of x^6: 21/1024)
// -- a polynomial in ONE symmetric matrix, so every product is symmetric (6 entries, 18 FMAs) and Horner needs four of them.  Measured in
// the block kernel: 66 VALU instructions per wave fewer than the Newton steps of polar_rotation plus (F - R) F^T (1263 -> 1197,
// profiles/r4u).  Truncation error at |E|_F = 0.12 (stretches within ~6 % of 1): 21/1024 x^6 = 6e-8 against x/2 = 0.06 -- 1e-6 relative,
// the float32 floor; and the stress is free of the F - R cancellation: 6e-8 of 2 mu from the float64 value under arbitrary rotations,
// where the rotation route has 2.8e-7 (tests/test_mpm_oracle.py).
// Valid only for det F > 0 and small E: the caller tests both per lane and takes the rotation route otherwise.
struct Sym3 {
    float xx, xy, xz, yy, yz, zz;
};
PX_HD Sym3 sym_mul_commuting(const Sym3& a, const Sym3& b) {   // a b for commuting symmetric a, b (the product is symmetric)
    Sym3 c;
    c.xx = a.xx * b.xx + a.xy * b.xy + a.xz * b.xz;
    c.xy = a.xx * b.xy + a.xy * b.yy + a.xz * b.yz;
    c.xz = a.xx * b.xz + a.xy * b.yz + a.xz * b.zz;
    c.yy = a.xy * b.xy + a.yy * b.yy + a.yz * b.yz;
    c.yz = a.xy * b.xz + a.yy * b.yz + a.yz * b.zz;
    c.zz = a.xz * b.xz + a.yz * b.yz + a.zz * b.zz;
    return c;
}
constexpr float kFcrSeriesMaxE2 = 0.0144f;   // |E|_F <= 0.12
PX_HD Sym3 fcr_strain(const Mat3& F) {       // E = F F^T - I
    // The diagonal starts from fma(F_ii, F_ii, -1): exact product, one rounding of a SMALL number when F is near the identity (so E
    // keeps full relative precision there, where it matters most), and no worse than 1e-7 absolute under a large rotation.
    Sym3 E;
    E.xx = fmaf(F.m[2], F.m[2], fmaf(F.m[1], F.m[1], fmaf(F.m[0], F.m[0], -1.0f)));
    E.yy = fmaf(F.m[5], F.m[5], fmaf(F.m[3], F.m[3], fmaf(F.m[4], F.m[4], -1.0f)));
    E.zz = fmaf(F.m[7], F.m[7], fmaf(F.m[6], F.m[6], fmaf(F.m[8], F.m[8], -1.0f)));
    E.xy = fmaf(F.m[2], F.m[5], fmaf(F.m[1], F.m[4], F.m[0] * F.m[3]));
    E.xz = fmaf(F.m[2], F.m[8], fmaf(F.m[1], F.m[7], F.m[0] * F.m[6]));
    E.yz = fmaf(F.m[5], F.m[8], fmaf(F.m[4], F.m[7], F.m[3] * F.m[6]));
    return E;
}
PX_HD float sym_norm2(const Sym3& E) {
    return E.xx * E.xx + E.yy * E.yy + E.zz * E.zz + 2.0f * (E.xy * E.xy + E.xz * E.xz + E.yz * E.yz);
}
PX_HD Sym3 fcr_b_minus_sqrt_b(const Sym3& E) {   // E (1/2 + E (1/8 + E (-1/16 + E (5/128 - 7/256 E))))
    Sym3 h;
    const float c5 = -7.0f / 256.0f, c4 = 5.0f / 128.0f, c3 = -1.0f / 16.0f, c2 = 0.125f, c1 = 0.5f;
    h.xx = c5 * E.xx + c4; h.yy = c5 * E.yy + c4; h.zz = c5 * E.zz + c4;
    h.xy = c5 * E.xy; h.xz = c5 * E.xz; h.yz = c5 * E.yz;
    h = sym_mul_commuting(E, h); h.xx += c3; h.yy += c3; h.zz += c3;
    h = sym_mul_commuting(E, h); h.xx += c2; h.yy += c2; h.zz += c2;
    h = sym_mul_commuting(E, h); h.xx += c1; h.yy += c1; h.zz += c1;
    return sym_mul_commuting(E, h);
}

// ---------------------------------------------------------------- constitutive models
struct MaterialScalars {  // MPMModelStruct scalars, warp_utils.py:24-36
    float alpha, hardening, xi, softening, plastic_viscosity;
};

PX_HD float vlen3(float a, float b, float c) { return px_sqrt(a * a + b * b + c * c); }

// kirchoff_stress_water, mpm_utils.py:20-28
PX_HD Mat3 stress_water(float J, float bulk) {
    const float pressure = -bulk * (powf(J, -1.1f) - 1.0f);
    Mat3 T;
    for (int i = 0; i < 9; ++i) T.m[i] = 0.0f;
    T.m[0] = T.m[4] = T.m[8] = J * pressure;
    return T;
}

// ---- everything that goes through the SVD: ONE decomposition per particle, and only its left half ---------------------
// The reference decomposes twice: F_trial inside the return mapping (mpm_utils.py:94,145,202,249) and the returned F again
// for the stress (:501).  The second one decomposes a matrix whose factors are already in hand:
//   * no yield:  F = F_trial, same U, sigma, V;
//   * yield:     F = U diag(sigma') V^T was just ASSEMBLED from proper rotations and positive sigma' (:115-124 etc.), so its
//     SVD is (U P, P^T sigma', V P) for a permutation P -- and every stress law below is invariant under P.
// Every stress of the SVD family has the form  P F^T  with  P = U diag(p) V^T  (:17 with R = U V^T, :63-68, :86), so with
// F^T = V diag(sigma) U^T it collapses to  tau = U diag(t) U^T,  t_d = p_d sigma_d:
//   fixed-corotated (0, 5):  t_d = 2 mu (sigma_d - 1) sigma_d + lam J (J - 1)
//   StVK-Hencky (1, 3):      t_d = (2 mu eps_d + lam tr eps) sigma_d,   eps_d = log max(sigma_d, 0.01)
//   Drucker-Prager (2):      t_d = (2 mu log sigma_d + lam tr log sigma) / sigma_d * sigma_d
// -- symmetric by construction (the reference's (T + T^T)/2 is the identity on it) and free of the F - R cancellation.
// The returned F needs no V either:  U diag(sigma') V^T = U diag(sigma'/sigma) U^T F_trial, written as the CORRECTION
//     F = F_trial + (U diag(sigma'_d/sigma_d - 1) U^T) F_trial,      sigma'_d/sigma_d = exp(eps'_d - eps_d)
// so a particle that yields a little moves F a little (the re-assembly from three factors re-rounds every entry of F at
// 1e-7 whether it moved or not).  So the decomposition is left_stretch: Jacobi on F F^T, U and sigma only.
// A wave whose lanes hold different plastic materials runs it once for all of them; only the three-scalar return maps
// diverge.  tests/test_mpm_oracle.py holds this against the tests' two-SVD restatement of the reference (itself pinned to
// the reference's code, tests/test_mpm_ref_golden.py).
constexpr float kLogSigMin = -4.605170186f;   // log(0.01): the clamp of :98,:149,:206 and :57 in log space

PX_HD bool is_svd_material(int material) { return material == 1 || material == 2 || material == 3 || material == 5; }

// one returned singular value: sigma' = exp(e_new) and sigma'/sigma - 1, from the log-space step e_new - e when e is the log
// of sigma itself (sigma >= floor > 0: one exp, the step keeps its relative accuracy), from the values otherwise (clamped
// or inverted sigma: the reference's exp of the clamped strain against the raw sigma)
PX_HD void returned_sigma(float so, float e, float e_new, float floor_, float& sn, float& rm1) {
    if (so >= floor_) {
        const float r = px_exp(e_new - e);
        rm1 = r - 1.0f;
        sn = so * r;
    } else {
        sn = px_exp(e_new);
        rm1 = px_div(sn, so) - 1.0f;
    }
}

// Return mapping of the singular values `so` of F_trial (signed, Warp convention; any order) for material 1 / 2 / 3 / 5 and
// the principal Kirchhoff stresses t of the returned F.  Returns true when F moved:  F = F_trial + U diag(rm1) U^T F_trial
// with singular values sn; false: F = F_trial (sn = so, rm1 = 0).  J_trial = det F_trial.
// mu / lam / ys are the particle's mutable model entries (snow damage, hardening).
PX_HD bool return_map_principal(int material, const float so[3], float J_trial, float& mu, float& lam, float& ys,
                                const MaterialScalars& ms, float dt, float sn[3], float rm1[3], float t[3]) {
    bool moved = false;
    for (int d = 0; d < 3; ++d) { sn[d] = so[d]; rm1[d] = 0.0f; }
    if (material == 2) {
        // sand_return_mapping, mpm_utils.py:242-279 + kirchoff_stress_drucker_prager, :71-86
        float e[3];
        for (int d = 0; d < 3; ++d) e[d] = px_log(fmaxf(fabsf(so[d]), 1e-14f));
        const float tr = e[0] + e[1] + e[2];
        const float eh[3] = {e[0] - tr * (1.0f / 3.0f), e[1] - tr * (1.0f / 3.0f), e[2] - tr * (1.0f / 3.0f)};
        const float ehn = vlen3(eh[0], eh[1], eh[2]);
        const float dg = ehn + px_div(3.0f * lam + 2.0f * mu, 2.0f * mu) * tr * ms.alpha;
        float l[3] = {e[0], e[1], e[2]};            // log of the returned F's singular values
        if (dg > 0.0f) {
            moved = true;
            const float k = (tr > 0.0f) ? 0.0f : px_div(dg, ehn);
            for (int d = 0; d < 3; ++d) {
                l[d] = (tr > 0.0f) ? 0.0f : e[d] - eh[d] * k;   // expansion: F = U V^T (sigma' = 1), stress-free
                returned_sigma(so[d], e[d], l[d], 1e-14f, sn[d], rm1[d]);
            }
        } else if (!(fminf(fminf(so[0], so[1]), so[2]) >= 1e-14f)) {
            // elastic with a collapsed / inverted element: the reference takes log(sigma) of the raw value (:75), NaN for < 0
            for (int d = 0; d < 3; ++d) l[d] = px_log(so[d]);
        }
        const float ltr = l[0] + l[1] + l[2];
        for (int d = 0; d < 3; ++d) t[d] = 2.0f * mu * l[d] + lam * ltr;
        return moved;
    }
    // 1 metal, 3 visco-plastic, 5 snow: Hencky strain of the clamped singular values (:98-101, :149-152, :206-211)
    float e[3], en[3];
    for (int d = 0; d < 3; ++d) e[d] = en[d] = px_log(fmaxf(so[d], 0.01f));
    const float tr = e[0] + e[1] + e[2];
    const float eh[3] = {e[0] - tr * (1.0f / 3.0f), e[1] - tr * (1.0f / 3.0f), e[2] - tr * (1.0f / 3.0f)};
    if (material == 3) {
        // viscoplasticity_return_mapping_with_StVK, :195-239
        const float two_mu = 2.0f * mu;
        const float sn_ = two_mu * vlen3(eh[0], eh[1], eh[2]);            // |s_trial|
        const float y = sn_ - 0.8164965809f * ys;
        if (y > 0.0f) {
            float b = 0.0f;
            for (int d = 0; d < 3; ++d) { const float sg = fmaxf(so[d], 0.01f); b += sg * sg; }
            const float mu_hat = mu * b * (1.0f / 3.0f);
            const float snn = sn_ - px_div(y, 1.0f + px_div(ms.plastic_viscosity, 2.0f * mu_hat * dt));
            for (int d = 0; d < 3; ++d) en[d] = px_rcp(two_mu) * (px_div(snn, sn_) * (two_mu * eh[d])) + tr * (1.0f / 3.0f);
            moved = true;
        }
    } else {
        // von_mises_return_mapping (:89-135) / ..._with_damage (:138-191)
        const bool damage = (material == 5);
        float tau[3];
        for (int d = 0; d < 3; ++d) tau[d] = 2.0f * mu * e[d] + lam * tr;
        const float st = tau[0] + tau[1] + tau[2];
        const float cn = vlen3(tau[0] - st * (1.0f / 3.0f), tau[1] - st * (1.0f / 3.0f), tau[2] - st * (1.0f / 3.0f));
        if (cn > ys && !(damage && ys <= 0.0f)) {
            const float ehn = vlen3(eh[0], eh[1], eh[2]) + 1e-6f;
            const float dg = ehn - px_div(ys, 2.0f * mu);
            const float k = px_div(dg, ehn);
            for (int d = 0; d < 3; ++d) en[d] = e[d] - k * eh[d];
            if (damage) {
                ys = ys - ms.softening * vlen3(k * eh[0], k * eh[1], k * eh[2]);
                if (ys <= 0.0f) { mu = 0.0f; lam = 0.0f; }
            }
            if (ms.hardening == 1.0f) ys = ys + 2.0f * mu * ms.xi * dg;
            moved = true;
        }
    }
    if (moved)
        for (int d = 0; d < 3; ++d) returned_sigma(so[d], e[d], en[d], 0.01f, sn[d], rm1[d]);
    if (material == 5) {
        // kirchoff_stress_FCR (:10-17) in principal form; J of the returned F
        const float J = moved ? sn[0] * sn[1] * sn[2] : J_trial;
        const float iso = lam * J * (J - 1.0f);
        for (int d = 0; d < 3; ++d) t[d] = 2.0f * mu * (sn[d] - 1.0f) * sn[d] + iso;
    } else {
        // kirchoff_stress_StVK (:52-68): the strain of the returned F, clamped as the stress function clamps it
        float ec[3];
        for (int d = 0; d < 3; ++d) ec[d] = moved ? fmaxf(en[d], kLogSigMin) : e[d];
        const float etr = ec[0] + ec[1] + ec[2];
        for (int d = 0; d < 3; ++d) t[d] = (2.0f * mu * ec[d] + lam * etr) * sn[d];
    }
    return moved;
}

// U diag(d) U^T, symmetric
PX_HD Mat3 mat_udut(const Mat3& U, const float d[3]) {
    Mat3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) {
            const float v = U.m[3 * i] * d[0] * U.m[3 * j] + U.m[3 * i + 1] * d[1] * U.m[3 * j + 1] + U.m[3 * i + 2] * d[2] * U.m[3 * j + 2];
            r.m[3 * i + j] = v;
            r.m[3 * j + i] = v;
        }
    return r;
}

// fixed-corotated stress by the decomposition route (inverted or badly conditioned jelly): principal form of mpm_utils.py:10-17
PX_HD Mat3 stress_fcr_svd(const Mat3& F, float J, float mu, float lam) {
    Mat3 U;
    float sg[3], t[3];
    left_stretch(F, J, U, sg);
    const float iso = lam * J * (J - 1.0f);
    for (int d = 0; d < 3; ++d) t[d] = 2.0f * mu * (sg[d] - 1.0f) * sg[d] + iso;
    return mat_udut(U, t);
}

// Stress of a GIVEN elastic F (the export path, mpm_solver_warp.py:725-741, and the jelly / water branch of the step):
// the constitutive half of compute_stress_from_F_trial (mpm_utils.py:495-526), symmetrised.  Material ids
// (mpm_solver_warp.py:10-18): 0 jelly, 1 metal, 2 sand, 3 visplas, 5 snow, 6 "stationary" -- which the reference treats as
// the water EOS with `bulk` (0 in every shipped flow => tau = 0); any other id gives tau = 0.
PX_HD Mat3 kirchhoff_stress(int material, const Mat3& F, float mu, float lam, float bulk) {
    const float J = mat_det(F);
    Mat3 T;
    for (int i = 0; i < 9; ++i) T.m[i] = 0.0f;
    Mat3 Rp;
    Sym3 E;
    bool series = false;
    if (material == 0) {
        E = fcr_strain(F);
        series = (J > 0.0f) && (sym_norm2(E) <= kFcrSeriesMaxE2);   // the lane's own decision (NaN: false)
    }
    if (series) {
        // fixed-corotated jelly at small strain: 2 mu (b - sqrt b) + lam J (J - 1) I, no rotation needed (see above)
        const Sym3 P = fcr_b_minus_sqrt_b(E);
        const float two_mu = 2.0f * mu, iso = lam * J * (J - 1.0f);
        T.m[0] = two_mu * P.xx + iso; T.m[4] = two_mu * P.yy + iso; T.m[8] = two_mu * P.zz + iso;
        T.m[1] = T.m[3] = two_mu * P.xy; T.m[2] = T.m[6] = two_mu * P.xz; T.m[5] = T.m[7] = two_mu * P.yz;
        return T;   // symmetric by construction
    } else if (material == 6) {
        return stress_water(J, bulk);   // isotropic
    } else if (material == 0 && polar_rotation(F, Rp)) {
        // fixed-corotated jelly: only the rotation is needed -- Newton polar instead of the decomposition
        Mat3 D;
        for (int i = 0; i < 9; ++i) D.m[i] = 2.0f * mu * (F.m[i] - Rp.m[i]);
        T = mat_mul_bt(D, F);
        const float iso = lam * J * (J - 1.0f);
        T.m[0] += iso; T.m[4] += iso; T.m[8] += iso;
    } else if (material == 0) {
        return stress_fcr_svd(F, J, mu, lam);
    } else if (is_svd_material(material)) {
        // the stress of an F that is already elastic: the stress half of return_map_principal with nothing to return
        Mat3 U;
        float sg[3], t[3];
        left_stretch(F, J, U, sg);
        if (material == 5) {
            const float iso = lam * J * (J - 1.0f);
            for (int d = 0; d < 3; ++d) t[d] = 2.0f * mu * (sg[d] - 1.0f) * sg[d] + iso;
        } else {
            float l[3];
            for (int d = 0; d < 3; ++d) l[d] = (material == 2) ? px_log(sg[d]) : px_log(fmaxf(sg[d], 0.01f));
            const float ltr = l[0] + l[1] + l[2];
            for (int d = 0; d < 3; ++d) t[d] = (2.0f * mu * l[d] + lam * ltr) * ((material == 2) ? 1.0f : sg[d]);
        }
        return mat_udut(U, t);
    }
    Mat3 tau;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) tau.m[3 * i + j] = (T.m[3 * i + j] + T.m[3 * j + i]) * 0.5f;
    return tau;
}

// A yielded F_trial with a singular value at float32's zero (|sigma_d| <= 1e-6 sigma_max, zero itself included): sigma'_d / sigma_d
// is infinite and the correction form F_trial + U diag(sigma'/sigma - 1) U^T F_trial turns the particle into NaN, which P2G spreads to
// its neighbours.  The reference re-assembles F = U diag(sigma') V^T, which stays finite -- the point of its max(sigma, 0.01)
// clamp -- so the same is done here: v_d = F_trial^T u_d / sigma_d on the regular axes, the null axes completed to a right-handed
// orthonormal frame (one null axis: the cross product of the other two, unique; more: any completion, as arbitrary as an SVD's own).
PX_HD void cross3(const float p[3], const float q[3], float w[3]) {
    w[0] = p[1] * q[2] - p[2] * q[1]; w[1] = p[2] * q[0] - p[0] * q[2]; w[2] = p[0] * q[1] - p[1] * q[0];
    const float rn = px_rsqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    w[0] *= rn; w[1] *= rn; w[2] *= rn;
}
struct Vec3s { float a, b, c; };
PX_RARE Mat3 rebuild_rank_deficient(Mat3 Ft, Mat3 U, Vec3s so_, Vec3s sn_, float amax) {
    const float so[3] = {so_.a, so_.b, so_.c}, sn[3] = {sn_.a, sn_.b, sn_.c};
    float v[3][3];
    bool reg[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        reg[d] = fabsf(so[d]) > 1.0e-6f * amax;
        const float inv = reg[d] ? 1.0f / so[d] : 0.0f;
        float n2 = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { v[d][c] = (Ft(0, c) * U(0, d) + Ft(1, c) * U(1, d) + Ft(2, c) * U(2, d)) * inv; n2 += v[d][c] * v[d][c]; }
        const float rn = reg[d] ? px_rsqrt(n2) : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[d][c] *= rn;
    }
    const int n_reg = (reg[0] ? 1 : 0) + (reg[1] ? 1 : 0) + (reg[2] ? 1 : 0);
#pragma unroll
    for (int a = 0; a < 3; ++a) {           // ONE regular axis a: a unit vector orthogonal to it (cross with the coordinate axis it is least aligned with)
        const int b = (a + 1) % 3;
        if (n_reg == 1 && reg[a]) {
            const float ax = fabsf(v[a][0]), ay = fabsf(v[a][1]), az = fabsf(v[a][2]);
            const bool ex = ax <= ay && ax <= az, ey = !ex && ay <= az;
            const float e[3] = {ex ? 1.0f : 0.0f, ey ? 1.0f : 0.0f, (!ex && !ey) ? 1.0f : 0.0f};
            cross3(v[a], e, v[b]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {           // (now two axes are known) the last one: v_d = v_{d+1} x v_{d+2}, right-handed with the other two
        const int d = (a + 2) % 3, b = (a + 1) % 3;      // d, a, b cyclic
        if (n_reg == 1 ? reg[a] : (n_reg == 2 && !reg[d])) cross3(v[a], v[b], v[d]);
    }
    Mat3 F;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // F_trial = 0 (or not finite): V = U
            const float v0 = n_reg ? v[0][c] : U(c, 0), v1 = n_reg ? v[1][c] : U(c, 1), v2 = n_reg ? v[2][c] : U(c, 2);
            F(r, c) = sn[0] * U(r, 0) * v0 + sn[1] * U(r, 1) * v1 + sn[2] * U(r, 2) * v2;
        }
    return F;
}

// compute_stress_from_F_trial, mpm_utils.py:467-526: F = returnMap(F_trial); tau = stress(F).
// mu/lam/ys are the particle's mutable model entries (snow damage and hardening write them).
PX_HD void return_map_and_stress(int material, const Mat3& Ft, float& mu, float& lam, float bulk, float& ys,
                                 const MaterialScalars& ms, float dt, Mat3& F, Mat3& tau) {
    F = Ft;
    if (is_svd_material(material)) {
        Mat3 U;
        float so[3], sn[3], rm1[3], t[3];
        const float J = mat_det(Ft);
        left_stretch(Ft, J, U, so);
        if (return_map_principal(material, so, J, mu, lam, ys, ms, dt, sn, rm1, t)) {
            const float amax = fmaxf(fmaxf(fabsf(so[0]), fabsf(so[1])), fabsf(so[2]));
            const bool rank_deficient = !(fabsf(so[0]) > 1.0e-6f * amax) || !(fabsf(so[1]) > 1.0e-6f * amax) || !(fabsf(so[2]) > 1.0e-6f * amax);
            if (rank_deficient) {
                F = rebuild_rank_deficient(Ft, U, Vec3s{so[0], so[1], so[2]}, Vec3s{sn[0], sn[1], sn[2]}, amax);
            } else {
                const Mat3 GF = mat_mul(mat_udut(U, rm1), Ft);
                for (int i = 0; i < 9; ++i) F.m[i] = Ft.m[i] + GF.m[i];
            }
        }
        tau = mat_udut(U, t);
    } else {
        tau = kirchhoff_stress(material, Ft, mu, lam, bulk);
    }
}

// ---------------------------------------------------------------- B-spline stencil
// Shared by P2G and G2P (mpm_utils.py:343-358, :418-434).  base = trunc(x*inv_dx - 0.5) (wp.int
// truncates toward zero), w[d][i] = weight of offset i along axis d, dw = its derivative in cell units.
struct Stencil {
    int base[3];
    float fx[3];
    float w[3][3];
    float dw[3][3];
};
PX_HD Stencil make_stencil(float x, float y, float z, float inv_dx) {
    Stencil s;
    const float p[3] = {x, y, z};
    for (int d = 0; d < 3; ++d) {
        const float gp = p[d] * inv_dx;
        s.base[d] = (int)(gp - 0.5f);
        s.fx[d] = gp - (float)s.base[d];
        const float wa = 1.5f - s.fx[d], wb = s.fx[d] - 1.0f, wc = s.fx[d] - 0.5f;
        s.w[d][0] = wa * wa * 0.5f;
        s.w[d][1] = 0.75f - wb * wb;
        s.w[d][2] = wc * wc * 0.5f;
        s.dw[d][0] = s.fx[d] - 1.5f;
        s.dw[d][1] = -2.0f * (s.fx[d] - 1.0f);
        s.dw[d][2] = s.fx[d] - 0.5f;
    }
    return s;
}

}  // namespace pixie
