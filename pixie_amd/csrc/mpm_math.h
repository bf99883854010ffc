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
#else
#define PX_HD inline
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

// One Jacobi rotation on the symmetric matrix S (upper storage s00,s01,s02,s11,s12,s22 as a
// full Mat3) annihilating S(p,q), accumulated into V.
template <int P, int Q>
PX_HD void jacobi_rotate(Mat3& S, Mat3& V) {
    const float apq = S(P, Q);
    const float app = S(P, P), aqq = S(Q, Q);
    // theta = (aqq-app)/(2apq); t = sgn(theta)/(|theta|+sqrt(theta^2+1)), written without the
    // division by apq so that apq -> 0 gives t -> 0 smoothly.
    const float d = aqq - app;
    const float two_apq = 2.0f * apq;
    const float h = sqrtf(d * d + two_apq * two_apq);
    // t = 2apq / (d + sign(d) h)
    const float denom = d + (d >= 0.0f ? h : -h);
    float t = (fabsf(denom) > 1e-30f) ? two_apq / denom : 0.0f;
    if (fabsf(apq) <= 1e-12f * (fabsf(app) + fabsf(aqq))) t = 0.0f;
    const float c = px_rsqrt(t * t + 1.0f);
    const float s = t * c;
    // S <- J^T S J with J = [[c, s],[-s, c]] on (P,Q)
    constexpr int R = 3 - P - Q;
    const float arp = S(R, P), arq = S(R, Q);
    S(P, P) = app - t * apq;
    S(Q, Q) = aqq + t * apq;
    S(P, Q) = 0.0f; S(Q, P) = 0.0f;
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
PX_HD void svd3(const Mat3& F, Mat3& U, float sig[3], Mat3& V) {
    Mat3 S;  // F^T F
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) S(i, j) = F(0, i) * F(0, j) + F(1, i) * F(1, j) + F(2, i) * F(2, j);
    V = mat_identity();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int sweep = 0; sweep < 5; ++sweep) {
        jacobi_rotate<0, 1>(S, V);
        jacobi_rotate<0, 2>(S, V);
        jacobi_rotate<1, 2>(S, V);
    }
    // 15 accumulated fp32 rotations leave V orthogonal only to ~1e-6; one Newton-Schulz step
    // V <- V (3I - V^T V)/2 squares that error, which keeps R = U V^T accurate to fp32 roundoff.
    {
        Mat3 G;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const float g = V(0, i) * V(0, j) + V(1, i) * V(1, j) + V(2, i) * V(2, j);
                G(i, j) = ((i == j) ? 1.5f : 0.0f) - 0.5f * g;
            }
        V = mat_mul(V, G);
    }
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

PX_HD float vlen3(float a, float b, float c) { return sqrtf(a * a + b * b + c * c); }

// kirchoff_stress_FCR, mpm_utils.py:10-17:  2 mu (F - R) F^T + lam J (J-1) I
PX_HD Mat3 stress_fcr(const Mat3& F, const Mat3& U, const Mat3& V, float J, float mu, float lam) {
    Mat3 R = mat_mul_bt(U, V);
    Mat3 D;
    for (int i = 0; i < 9; ++i) D.m[i] = 2.0f * mu * (F.m[i] - R.m[i]);
    Mat3 T = mat_mul_bt(D, F);
    const float iso = lam * J * (J - 1.0f);
    T.m[0] += iso; T.m[4] += iso; T.m[8] += iso;
    return T;
}
// kirchoff_stress_StVK, mpm_utils.py:52-68
PX_HD Mat3 stress_stvk(const Mat3& F, const Mat3& U, const Mat3& V, const float sig_in[3], float mu, float lam) {
    float e[3], tau[3];
    for (int d = 0; d < 3; ++d) e[d] = logf(fmaxf(sig_in[d], 0.01f));
    const float tr = e[0] + e[1] + e[2];
    for (int d = 0; d < 3; ++d) tau[d] = 2.0f * mu * e[d] + lam * tr;
    return mat_mul_bt(mat_udvt(U, tau, V), F);
}
// kirchoff_stress_drucker_prager, mpm_utils.py:71-86
PX_HD Mat3 stress_dp(const Mat3& F, const Mat3& U, const Mat3& V, const float sig[3], float mu, float lam) {
    float l[3], c[3];
    for (int d = 0; d < 3; ++d) l[d] = logf(sig[d]);
    const float tr = l[0] + l[1] + l[2];
    for (int d = 0; d < 3; ++d) {
        const float inv = 1.0f / sig[d];
        c[d] = 2.0f * mu * l[d] * inv + lam * tr * inv;
    }
    return mat_mul_bt(mat_udvt(U, c, V), F);
}
// kirchoff_stress_water, mpm_utils.py:20-28
PX_HD Mat3 stress_water(float J, float bulk) {
    const float pressure = -bulk * (powf(J, -1.1f) - 1.0f);
    Mat3 T;
    for (int i = 0; i < 9; ++i) T.m[i] = 0.0f;
    T.m[0] = T.m[4] = T.m[8] = J * pressure;
    return T;
}

// von_mises_return_mapping (mpm_utils.py:89-135) and ..._with_damage (:138-191).
// ys/mu/lam are the particle's mutable model entries.
PX_HD Mat3 rm_von_mises(const Mat3& Ft, float& ys, float& mu, float& lam, const MaterialScalars& ms, bool damage) {
    Mat3 U, V;
    float so[3];
    svd3(Ft, U, so, V);
    float e[3];
    for (int d = 0; d < 3; ++d) e[d] = logf(fmaxf(so[d], 0.01f));
    const float tr = e[0] + e[1] + e[2];
    const float temp = tr / 3.0f;
    float tau[3];
    for (int d = 0; d < 3; ++d) tau[d] = 2.0f * mu * e[d] + lam * tr;
    const float st = tau[0] + tau[1] + tau[2];
    const float cn = vlen3(tau[0] - st / 3.0f, tau[1] - st / 3.0f, tau[2] - st / 3.0f);
    if (cn > ys) {
        if (damage && ys <= 0.0f) return Ft;
        const float eh[3] = {e[0] - temp, e[1] - temp, e[2] - temp};
        const float ehn = vlen3(eh[0], eh[1], eh[2]) + 1e-6f;
        const float dg = ehn - ys / (2.0f * mu);
        const float k = dg / ehn;
        float ex[3];
        for (int d = 0; d < 3; ++d) ex[d] = expf(e[d] - k * eh[d]);
        if (damage) {
            ys = ys - ms.softening * vlen3(k * eh[0], k * eh[1], k * eh[2]);
            if (ys <= 0.0f) { mu = 0.0f; lam = 0.0f; }
        }
        Mat3 Fe = mat_udvt(U, ex, V);
        if (ms.hardening == 1.0f) ys = ys + 2.0f * mu * ms.xi * dg;
        return Fe;
    }
    return Ft;
}
// viscoplasticity_return_mapping_with_StVK, mpm_utils.py:195-239
PX_HD Mat3 rm_visco(const Mat3& Ft, float ys, float mu, const MaterialScalars& ms, float dt) {
    Mat3 U, V;
    float so[3], sg[3], e[3];
    svd3(Ft, U, so, V);
    for (int d = 0; d < 3; ++d) { sg[d] = fmaxf(so[d], 0.01f); e[d] = logf(sg[d]); }
    const float tr = e[0] + e[1] + e[2];
    const float eh[3] = {e[0] - tr / 3.0f, e[1] - tr / 3.0f, e[2] - tr / 3.0f};
    const float s[3] = {2.0f * mu * eh[0], 2.0f * mu * eh[1], 2.0f * mu * eh[2]};
    const float sn = vlen3(s[0], s[1], s[2]);
    const float y = sn - sqrtf(2.0f / 3.0f) * ys;
    if (y > 0.0f) {
        const float mu_hat = mu * (sg[0] * sg[0] + sg[1] * sg[1] + sg[2] * sg[2]) / 3.0f;
        const float snn = sn - y / (1.0f + ms.plastic_viscosity / (2.0f * mu_hat * dt));
        float ex[3];
        for (int d = 0; d < 3; ++d) ex[d] = expf(1.0f / (2.0f * mu) * ((snn / sn) * s[d]) + tr / 3.0f);
        return mat_udvt(U, ex, V);
    }
    return Ft;
}
// sand_return_mapping, mpm_utils.py:242-279
PX_HD Mat3 rm_sand(const Mat3& Ft, float mu, float lam, const MaterialScalars& ms) {
    Mat3 U, V;
    float sg[3], e[3];
    svd3(Ft, U, sg, V);
    for (int d = 0; d < 3; ++d) e[d] = logf(fmaxf(fabsf(sg[d]), 1e-14f));
    const float tr = e[0] + e[1] + e[2];
    const float eh[3] = {e[0] - tr / 3.0f, e[1] - tr / 3.0f, e[2] - tr / 3.0f};
    const float ehn = vlen3(eh[0], eh[1], eh[2]);
    const float dg = ehn + (3.0f * lam + 2.0f * mu) / (2.0f * mu) * tr * ms.alpha;
    if (dg <= 0.0f) return Ft;
    if (tr > 0.0f) return mat_mul_bt(U, V);
    float ex[3];
    for (int d = 0; d < 3; ++d) ex[d] = expf(e[d] - eh[d] * (dg / ehn));
    return mat_udvt(U, ex, V);
}

// Constitutive half of compute_stress_from_F_trial (mpm_utils.py:495-526): Kirchhoff stress of the
// (already return-mapped) F, symmetrised.  Material ids (mpm_solver_warp.py:10-18): 0 jelly, 1 metal,
// 2 sand, 3 visplas, 5 snow, 6 "stationary" -- which the reference treats as the water EOS with `bulk`
// (0 in every shipped flow => tau = 0); any other id gives tau = 0.
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
        T = stress_water(J, bulk);
    } else if (material == 0 && polar_rotation(F, Rp)) {
        // fixed-corotated jelly: only the rotation is needed -- Newton polar instead of the full SVD
        Mat3 D;
        for (int i = 0; i < 9; ++i) D.m[i] = 2.0f * mu * (F.m[i] - Rp.m[i]);
        T = mat_mul_bt(D, F);
        const float iso = lam * J * (J - 1.0f);
        T.m[0] += iso; T.m[4] += iso; T.m[8] += iso;
    } else if (material == 0 || material == 5 || material == 1 || material == 2 || material == 3) {
        Mat3 U, V;
        float sg[3];
        svd3(F, U, sg, V);
        if (material == 0 || material == 5) T = stress_fcr(F, U, V, J, mu, lam);
        else if (material == 2) T = stress_dp(F, U, V, sg, mu, lam);
        else T = stress_stvk(F, U, V, sg, mu, lam);
    }
    Mat3 tau;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) tau.m[3 * i + j] = (T.m[3 * i + j] + T.m[3 * j + i]) * 0.5f;
    return tau;
}

// compute_stress_from_F_trial, mpm_utils.py:467-526: F = returnMap(F_trial); tau = stress(F).
// mu/lam/ys are the particle's mutable model entries (snow damage and hardening write them).
PX_HD void return_map_and_stress(int material, const Mat3& Ft, float& mu, float& lam, float bulk, float& ys,
                                 const MaterialScalars& ms, float dt, Mat3& F, Mat3& tau) {
    if (material == 1) F = rm_von_mises(Ft, ys, mu, lam, ms, false);
    else if (material == 2) F = rm_sand(Ft, mu, lam, ms);
    else if (material == 3) F = rm_visco(Ft, ys, mu, ms, dt);
    else if (material == 5) F = rm_von_mises(Ft, ys, mu, lam, ms, true);
    else F = Ft;
    tau = kirchhoff_stress(material, F, mu, lam, bulk);
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
