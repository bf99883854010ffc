/*
 * oracle/mpm_oracle.c -- TEST INFRASTRUCTURE ONLY (parity checker + cpu_baseline leg).
 *
 * CPU restatement, in plain C, of the PhysGaussian MLS-MPM substep that the
 * reference runs as NVIDIA Warp 0.10.1 kernels.  Nothing in pixie_amd/ (the
 * product) may link, import or execute this file; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, and only as the checker.
 *
 * PINNED to the reference's own code (round 4): tests/golden/mpm_ref_golden.npz holds rollouts computed by the
 * reference's mpm_solver_warp.py / mpm_utils.py / warp_utils.py themselves, imported unmodified on a numpy
 * interpreter of the Warp API subset they use (tests/golden/wp_shim, tests/golden/make_mpm_ref_golden.py): all
 * material ids, every BC type, every particle modifier, APIC / RPIC / PIC, inverted elements.  The float64 build of
 * this file reproduces every particle and grid field of those runs to <= 2e-14 (tests/test_mpm_ref_golden.py).
 * What stays a stand-in on BOTH sides is `wp.svd3`: it lives in Warp's native library (warp_lang==0.10.1, reference
 * setup.py:19), not in the reference tree.  Here: a float64 one-sided Jacobi SVD canonicalised to the convention of
 * Warp's implementation (U, V proper rotations, sign carried by the last singular value); the fixture uses LAPACK
 * canonicalised the same way and also records what LAPACK's own convention would give for det F < 0.
 *
 * Every function cites the reference lines it follows.  Paths are relative to
 * /root/reference/third_party/PhysGaussian/mpm_solver_warp/.
 *
 * Build twice:  gcc -O2 -ffp-contract=off -DREAL=float  ... -o libmpm_oracle_f32.so
 *               gcc -O2 -ffp-contract=off -DREAL=double ... -o libmpm_oracle_f64.so
 * Storage mirrors warp_utils.py:6-74: vec3 = 3 reals, mat33 = 9 reals row-major,
 * grids indexed [x][y][z] with z fastest.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
/* Only the third build (Makefile: libmpm_oracle_f32_omp.so, -fopenmp), used by bench.py's cpu_baseline leg, is
 * multi-threaded: the pragmas below are no-ops in the two builds the tests use, whose sums keep the serial order. */
#ifdef _OPENMP
#define OMP_FOR _Pragma("omp parallel for schedule(static)")
#define OMP_ATOMIC _Pragma("omp atomic")
#else
#define OMP_FOR
#define OMP_ATOMIC
#endif
typedef REAL real;

/* Literals inside the reference's kernels are float32 constants, and every scalar that reaches a kernel -- struct members
 * (dx, inv_dx, gravity, model scalars, collider / modifier parameters) and launch arguments (time, dt) -- is a float32.
 * Both builds therefore see the SAME float32-rounded problem data; the float64 build only does the arithmetic (and keeps
 * the particle / grid state) in double.  R_ = kernel literal, P_ = host value as the float32 the kernel receives. */
#define R_(x) ((real)(float)(x))
#define P_(x) ((real)(float)(x))

static inline real r_log(real x) { return sizeof(real) == 4 ? (real)logf((float)x) : (real)log((double)x); }
static inline real r_exp(real x) { return sizeof(real) == 4 ? (real)expf((float)x) : (real)exp((double)x); }
static inline real r_sqrt(real x) { return sizeof(real) == 4 ? (real)sqrtf((float)x) : (real)sqrt((double)x); }
static inline real r_pow(real x, real y) { return sizeof(real) == 4 ? (real)powf((float)x, (float)y) : (real)pow((double)x, (double)y); }
static inline real r_abs(real x) { return x < 0 ? -x : x; }
static inline real r_max(real a, real b) { return a > b ? a : b; }
static inline real r_min(real a, real b) { return a < b ? a : b; }

/* ------------------------------------------------------------------ mat33 */
static void m_mul(const real *A, const real *B, real *C) { /* C = A*B */
    real t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, t, sizeof t);
}
static void m_T(const real *A, real *B) {
    real t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * j + i];
    memcpy(B, t, sizeof t);
}
static void m_vec(const real *A, const real *v, real *o) { /* o = A*v */
    real t[3];
    for (int i = 0; i < 3; ++i) t[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static real m_det(const real *A) {
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
           A[2] * (A[3] * A[7] - A[4] * A[6]);
}
static void m_diag(real a, real b, real c, real *D) {
    memset(D, 0, 9 * sizeof(real));
    D[0] = a; D[4] = b; D[8] = c;
}
static real v_len(const real *v) { return r_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* ------------------------------------------------------------------ svd3
 * Stand-in for Warp's native wp.svd3 (called at mpm_utils.py:94,145,202,249,501,566).
 * float64 one-sided (Hestenes) Jacobi; canonical form = Warp's convention:
 *   A = U diag(s) V^T, det(U) = det(V) = +1, |s0| >= |s1| >= |s2|, only s2 may be < 0.
 */
static void svd3_f64(const double *A, double *U, double *S, double *V) {
    double B[9];
    memcpy(B, A, sizeof B);
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; ++i) {
                    al += B[3 * i + p] * B[3 * i + p];
                    be += B[3 * i + q] * B[3 * i + q];
                    ga += B[3 * i + p] * B[3 * i + q];
                }
                if (ga == 0.0 || fabs(ga) <= 1e-300) continue;
                double lim = 1e-17 * sqrt(al * be);
                if (fabs(ga) <= lim) continue;
                off += fabs(ga);
                double zeta = (be - al) / (2.0 * ga);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; ++i) {
                    double bp = B[3 * i + p], bq = B[3 * i + q];
                    B[3 * i + p] = c * bp - s * bq;
                    B[3 * i + q] = s * bp + c * bq;
                    double vp = V[3 * i + p], vq = V[3 * i + q];
                    V[3 * i + p] = c * vp - s * vq;
                    V[3 * i + q] = s * vp + c * vq;
                }
            }
        if (off == 0.0) break;
    }
    double n[3];
    for (int j = 0; j < 3; ++j)
        n[j] = sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
    /* sort columns by decreasing norm */
    int idx[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (n[idx[b]] > n[idx[a]]) { int t = idx[a]; idx[a] = idx[b]; idx[b] = t; }
    double Bs[9], Vs[9];
    for (int j = 0; j < 3; ++j) {
        S[j] = n[idx[j]];
        for (int i = 0; i < 3; ++i) { Bs[3 * i + j] = B[3 * i + idx[j]]; Vs[3 * i + j] = V[3 * i + idx[j]]; }
    }
    memcpy(V, Vs, sizeof Vs);
    /* U columns = normalised B columns; rank-deficient columns completed by cross products */
    for (int j = 0; j < 3; ++j) {
        if (S[j] > 1e-300 && S[j] > 1e-14 * S[0]) {
            for (int i = 0; i < 3; ++i) U[3 * i + j] = Bs[3 * i + j] / S[j];
        } else {
            S[j] = (S[j] > 1e-300) ? S[j] : 0.0;
            for (int i = 0; i < 3; ++i) U[3 * i + j] = 0.0;
        }
    }
    /* complete basis if needed (only for singular A) */
    double c0 = U[0] * U[0] + U[3] * U[3] + U[6] * U[6];
    if (c0 < 0.5) { U[0] = 1; U[3] = 0; U[6] = 0; }
    double c1 = U[1] * U[1] + U[4] * U[4] + U[7] * U[7];
    if (c1 < 0.5) {
        double a[3] = {U[0], U[3], U[6]};
        double e[3] = {0, 0, 0};
        int k = (fabs(a[0]) <= fabs(a[1]) && fabs(a[0]) <= fabs(a[2])) ? 0 : (fabs(a[1]) <= fabs(a[2]) ? 1 : 2);
        e[k] = 1.0;
        double d = a[k];
        double w[3] = {e[0] - d * a[0], e[1] - d * a[1], e[2] - d * a[2]};
        double wn = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        U[1] = w[0] / wn; U[4] = w[1] / wn; U[7] = w[2] / wn;
    }
    double c2 = U[2] * U[2] + U[5] * U[5] + U[8] * U[8];
    if (c2 < 0.5) {
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
    /* canonicalise to proper rotations, sign on the last singular value */
    double dU = U[0] * (U[4] * U[8] - U[5] * U[7]) - U[1] * (U[3] * U[8] - U[5] * U[6]) + U[2] * (U[3] * U[7] - U[4] * U[6]);
    if (dU < 0) { U[2] = -U[2]; U[5] = -U[5]; U[8] = -U[8]; S[2] = -S[2]; }
    double dV = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
    if (dV < 0) { V[2] = -V[2]; V[5] = -V[5]; V[8] = -V[8]; S[2] = -S[2]; }
}

/* The same algorithm carried out in float: what the float32 build uses, so that its distance from the float64 build
 * includes the rounding noise of a single-precision SVD (Warp's wp.svd3 runs in float32 on float32 matrices).  An
 * earlier version computed the float32 build's SVD in double and rounded the result, which made the "float32 oracle"
 * 2-3x more accurate in the plastic stresses than any float32 implementation of the reference can be. */
static void svd3_f32(const float *A, float *U, float *S, float *V) {
    float B[9];
    memcpy(B, A, sizeof B);
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    for (int sweep = 0; sweep < 60; ++sweep) {
        float off = 0.0f;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                float al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; ++i) {
                    al += B[3 * i + p] * B[3 * i + p];
                    be += B[3 * i + q] * B[3 * i + q];
                    ga += B[3 * i + p] * B[3 * i + q];
                }
                if (ga == 0.0f || fabsf(ga) <= 1e-37f) continue;
                float lim = 1e-8f * sqrtf(al * be);
                if (fabsf(ga) <= lim) continue;
                off += fabsf(ga);
                float zeta = (be - al) / (2.0f * ga);
                float t = (zeta >= 0 ? 1.0f : -1.0f) / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
                float c = 1.0f / sqrtf(1.0f + t * t), s = c * t;
                for (int i = 0; i < 3; ++i) {
                    float bp = B[3 * i + p], bq = B[3 * i + q];
                    B[3 * i + p] = c * bp - s * bq;
                    B[3 * i + q] = s * bp + c * bq;
                    float vp = V[3 * i + p], vq = V[3 * i + q];
                    V[3 * i + p] = c * vp - s * vq;
                    V[3 * i + q] = s * vp + c * vq;
                }
            }
        if (off == 0.0f) break;
    }
    float n[3];
    for (int j = 0; j < 3; ++j)
        n[j] = sqrtf(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
    /* sort columns by decreasing norm */
    int idx[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (n[idx[b]] > n[idx[a]]) { int t = idx[a]; idx[a] = idx[b]; idx[b] = t; }
    float Bs[9], Vs[9];
    for (int j = 0; j < 3; ++j) {
        S[j] = n[idx[j]];
        for (int i = 0; i < 3; ++i) { Bs[3 * i + j] = B[3 * i + idx[j]]; Vs[3 * i + j] = V[3 * i + idx[j]]; }
    }
    memcpy(V, Vs, sizeof Vs);
    /* U columns = normalised B columns; rank-deficient columns completed by cross products */
    for (int j = 0; j < 3; ++j) {
        if (S[j] > 1e-37f && S[j] > 1e-6f * S[0]) {
            for (int i = 0; i < 3; ++i) U[3 * i + j] = Bs[3 * i + j] / S[j];
        } else {
            S[j] = (S[j] > 1e-37f) ? S[j] : 0.0f;
            for (int i = 0; i < 3; ++i) U[3 * i + j] = 0.0f;
        }
    }
    /* complete basis if needed (only for singular A) */
    float c0 = U[0] * U[0] + U[3] * U[3] + U[6] * U[6];
    if (c0 < 0.5f) { U[0] = 1; U[3] = 0; U[6] = 0; }
    float c1 = U[1] * U[1] + U[4] * U[4] + U[7] * U[7];
    if (c1 < 0.5f) {
        float a[3] = {U[0], U[3], U[6]};
        float e[3] = {0, 0, 0};
        int k = (fabsf(a[0]) <= fabsf(a[1]) && fabsf(a[0]) <= fabsf(a[2])) ? 0 : (fabsf(a[1]) <= fabsf(a[2]) ? 1 : 2);
        e[k] = 1.0f;
        float d = a[k];
        float w[3] = {e[0] - d * a[0], e[1] - d * a[1], e[2] - d * a[2]};
        float wn = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        U[1] = w[0] / wn; U[4] = w[1] / wn; U[7] = w[2] / wn;
    }
    float c2 = U[2] * U[2] + U[5] * U[5] + U[8] * U[8];
    if (c2 < 0.5f) {
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
    /* canonicalise to proper rotations, sign on the last singular value */
    float dU = U[0] * (U[4] * U[8] - U[5] * U[7]) - U[1] * (U[3] * U[8] - U[5] * U[6]) + U[2] * (U[3] * U[7] - U[4] * U[6]);
    if (dU < 0) { U[2] = -U[2]; U[5] = -U[5]; U[8] = -U[8]; S[2] = -S[2]; }
    float dV = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
    if (dV < 0) { V[2] = -V[2]; V[5] = -V[5]; V[8] = -V[8]; S[2] = -S[2]; }
}

static void svd3(const real *A, real *U, real *S, real *V) {
    if (sizeof(real) == 4) {
        float a[9], u[9], s[3], v[9];
        for (int i = 0; i < 9; ++i) a[i] = (float)A[i];
        svd3_f32(a, u, s, v);
        for (int i = 0; i < 9; ++i) { U[i] = (real)u[i]; V[i] = (real)v[i]; }
        for (int i = 0; i < 3; ++i) S[i] = (real)s[i];
        return;
    }
    double a[9], u[9], s[3], v[9];
    for (int i = 0; i < 9; ++i) a[i] = (double)A[i];
    svd3_f64(a, u, s, v);
    for (int i = 0; i < 9; ++i) { U[i] = (real)u[i]; V[i] = (real)v[i]; }
    for (int i = 0; i < 3; ++i) S[i] = (real)s[i];
}

/* exported for tests */
void oracle_svd3(const real *A, real *U, real *S, real *V) { svd3(A, U, S, V); }

/* ------------------------------------------------------------------ state */
enum { BC_SURFACE = 0, BC_CUBOID = 1, BC_BBOX = 2 };
enum { PM_IMPULSE = 0, PM_TRANSLATION = 1, PM_ROTATION = 2 };

typedef struct {
    int type;
    real point[3], size[3], velocity[3], normal[3];
    real start_time, end_time, friction;
    int surface_type, reset;
} BC; /* Dirichlet_collider, warp_utils.py:78-109 */

typedef struct {
    int type;
    real point[3], size[3], force[3], velocity[3], normal[3], h1[3], h2[3];
    real half_height, radius, rotation_scale, translation_scale;
    real start_time, end_time;
    int *mask;
} PMod; /* Impulse_modifier / ParticleVelocityModifier, warp_utils.py:112-183 */

typedef struct {
    int n, ng;
    real grid_lim, dx, inv_dx;
    /* MPMStateStruct, warp_utils.py:42-74 */
    real *x, *v, *F, *F_trial, *C, *stress, *vol, *mass, *density;
#ifdef ORACLE_EXPERIMENT   /* tests/golden/attribute_config3_drift.py only: bit 0 = positions, bit 1 = deformation gradients carried in double */
    double *xd, *Fd, *Ftd;
    int exp_init;
#endif
    int *material, *selection;
    real *grid_m, *grid_v_in, *grid_v_out;
    /* MPMModelStruct, warp_utils.py:6-39 */
    real *E, *nu, *mu, *lam, *bulk, *yield_stress;
    real g[3], rpic_damping, grid_v_damping_scale, alpha, hardening, xi, softening, plastic_viscosity;
    double time;
    BC *bcs; int n_bc;
    PMod *pmods; int n_pmod;
    long oob; /* particles whose 3x3x3 stencil left the grid (UB in the reference) */
} MPM;

#define GI(s, ix, iy, iz) ((((size_t)(ix)) * (s)->ng + (iy)) * (s)->ng + (iz))

/* MPM_Simulator_WARP.initialize, mpm_solver_warp.py:52-180 */
/* mpm_solver_warp.py:84-86, 391-393: `wp.sin` / `wp.sqrt` called from Python scope run Warp's float32 built-ins
 * (float32 argument and result), the products around them are Python doubles, the struct member is a float32.  The
 * last bit of sinf is the host libm's; tests/golden/mpm_ref_golden.npz records the value numpy's float32 sin gives. */
static real host_alpha(double friction_angle) {
    double sin_phi = (double)sinf((float)(friction_angle / 180.0 * 3.14159265));
    return P_(((double)sqrtf((float)(2.0 / 3.0)) * 2.0 * sin_phi / (3.0 - sin_phi)));
}

MPM *mpm_create(int n, int n_grid, double grid_lim) {
    MPM *s = (MPM *)calloc(1, sizeof(MPM));
    s->n = n; s->ng = n_grid;
    s->grid_lim = P_(grid_lim);
    s->dx = P_((grid_lim / n_grid));           /* :62-66 */
    s->inv_dx = P_(((double)n_grid / grid_lim));
    size_t N = (size_t)n, G = (size_t)n_grid * n_grid * n_grid;
    s->x = calloc(3 * N, sizeof(real)); s->v = calloc(3 * N, sizeof(real));
    s->F = calloc(9 * N, sizeof(real)); s->F_trial = calloc(9 * N, sizeof(real));
    s->C = calloc(9 * N, sizeof(real)); s->stress = calloc(9 * N, sizeof(real));
    s->vol = calloc(N, sizeof(real)); s->mass = calloc(N, sizeof(real)); s->density = calloc(N, sizeof(real));
    s->material = calloc(N, sizeof(int)); s->selection = calloc(N, sizeof(int));
    s->E = calloc(N, sizeof(real)); s->nu = calloc(N, sizeof(real)); s->mu = calloc(N, sizeof(real));
    s->lam = calloc(N, sizeof(real)); s->bulk = calloc(N, sizeof(real)); s->yield_stress = calloc(N, sizeof(real));
    s->grid_m = calloc(G, sizeof(real)); s->grid_v_in = calloc(3 * G, sizeof(real)); s->grid_v_out = calloc(3 * G, sizeof(real));
    for (size_t p = 0; p < N; ++p) { s->F_trial[9 * p] = 1; s->F_trial[9 * p + 4] = 1; s->F_trial[9 * p + 8] = 1; } /* :272-277 */
    s->rpic_damping = 0; s->grid_v_damping_scale = R_(1.1); s->softening = R_(0.1); /* :79-92 */
    s->alpha = host_alpha(25.0);                                                      /* :84-86 */
    s->time = 0.0;
    return s;
}

void mpm_destroy(MPM *s) {
    if (!s) return;
    free(s->x); free(s->v); free(s->F); free(s->F_trial); free(s->C); free(s->stress);
    free(s->vol); free(s->mass); free(s->density); free(s->material); free(s->selection);
    free(s->E); free(s->nu); free(s->mu); free(s->lam); free(s->bulk); free(s->yield_stress);
    free(s->grid_m); free(s->grid_v_in); free(s->grid_v_out);
    for (int k = 0; k < s->n_pmod; ++k) free(s->pmods[k].mask);
    free(s->bcs); free(s->pmods); free(s);
}

/* field access by name: returns pointer + element count (in reals / ints) */
void *mpm_field(MPM *s, const char *name, long *count, int *is_int) {
    size_t N = (size_t)s->n, G = (size_t)s->ng * s->ng * s->ng;
    *is_int = 0;
#define FLD(nm, ptr, cnt) if (!strcmp(name, nm)) { *count = (long)(cnt); return (void *)(ptr); }
    FLD("x", s->x, 3 * N) FLD("v", s->v, 3 * N) FLD("F", s->F, 9 * N) FLD("F_trial", s->F_trial, 9 * N)
    FLD("C", s->C, 9 * N) FLD("stress", s->stress, 9 * N) FLD("vol", s->vol, N) FLD("mass", s->mass, N)
    FLD("density", s->density, N) FLD("E", s->E, N) FLD("nu", s->nu, N) FLD("mu", s->mu, N) FLD("lam", s->lam, N)
    FLD("bulk", s->bulk, N) FLD("yield_stress", s->yield_stress, N)
    FLD("grid_m", s->grid_m, G) FLD("grid_v_in", s->grid_v_in, 3 * G) FLD("grid_v_out", s->grid_v_out, 3 * G)
    *is_int = 1;
    FLD("material", s->material, N) FLD("selection", s->selection, N)
#undef FLD
    *count = 0;
    return NULL;
}

void mpm_set_scalar(MPM *s, const char *name, double val) {
    if (!strcmp(name, "rpic_damping")) s->rpic_damping = P_(val);
    else if (!strcmp(name, "grid_v_damping_scale")) s->grid_v_damping_scale = P_(val);
    else if (!strcmp(name, "hardening")) s->hardening = P_(val);
    else if (!strcmp(name, "xi")) s->xi = P_(val);
    else if (!strcmp(name, "softening")) s->softening = P_(val);
    else if (!strcmp(name, "plastic_viscosity")) s->plastic_viscosity = P_(val);
    else if (!strcmp(name, "gx")) s->g[0] = P_(val);
    else if (!strcmp(name, "gy")) s->g[1] = P_(val);
    else if (!strcmp(name, "gz")) s->g[2] = P_(val);
    else if (!strcmp(name, "time")) s->time = val;
    else if (!strcmp(name, "friction_angle")) s->alpha = host_alpha(val); /* mpm_solver_warp.py:390-393 */
    else if (!strcmp(name, "alpha")) s->alpha = P_(val);                  /* test hook: the value a fixture recorded */
}
double mpm_get_time(MPM *s) { return s->time; }
long mpm_get_oob(MPM *s) { return s->oob; }

/* get_float_array_product, warp_utils.py:234-241 (mass = density * vol) */
void mpm_update_mass(MPM *s) { for (int p = 0; p < s->n; ++p) s->mass[p] = s->density[p] * s->vol[p]; }

/* compute_mu_lam_from_E_nu, mpm_utils.py:282-288 */
void mpm_finalize_mu_lam(MPM *s) {
    for (int p = 0; p < s->n; ++p) {
        s->mu[p] = s->E[p] / (R_(2.0) * (R_(1.0) + s->nu[p]));
        s->lam[p] = s->E[p] * s->nu[p] / ((R_(1.0) + s->nu[p]) * (R_(1.0) - R_(2.0) * s->nu[p]));
    }
}
/* compute_bulk, mpm_utils.py:290-293 */
void mpm_compute_bulk(MPM *s) {
    for (int p = 0; p < s->n; ++p) s->bulk[p] = s->lam[p] + R_(2.) / R_(3.) * s->mu[p];
}

/* apply_additional_params, mpm_utils.py:591-610 */
void mpm_apply_additional_params(MPM *s, const double *point, const double *size, double E, double nu, double density, int material) {
    real pt[3], sz[3];
    for (int d = 0; d < 3; ++d) { pt[d] = P_(point[d]); sz[d] = P_(size[d]); }
    for (int p = 0; p < s->n; ++p) {
        const real *pos = s->x + 3 * p;
        if (pos[0] > pt[0] - sz[0] && pos[0] < pt[0] + sz[0] && pos[1] > pt[1] - sz[1] && pos[1] < pt[1] + sz[1] &&
            pos[2] > pt[2] - sz[2] && pos[2] < pt[2] + sz[2]) {
            s->E[p] = P_(E); s->nu[p] = P_(nu); s->density[p] = P_(density); s->material[p] = material;
        }
    }
}

/* ---- BC / modifier registration ------------------------------------------------ */
static BC *new_bc(MPM *s) {
    s->bcs = (BC *)realloc(s->bcs, (size_t)(s->n_bc + 1) * sizeof(BC));
    BC *b = &s->bcs[s->n_bc++];
    memset(b, 0, sizeof *b);
    return b;
}
/* add_surface_collider, mpm_solver_warp.py:749-783 */
void mpm_add_surface_collider(MPM *s, const double *point, const double *normal, int surface_type, double friction, double t0, double t1) {
    BC *b = new_bc(s);
    b->type = BC_SURFACE;
    double nn = 1.0 / sqrt(normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2]);
    for (int d = 0; d < 3; ++d) { b->point[d] = P_(point[d]); b->normal[d] = P_((nn * normal[d])); }
    b->surface_type = surface_type; b->friction = P_(friction); b->start_time = P_(t0); b->end_time = P_(t1);
}
/* set_velocity_on_cuboid, mpm_solver_warp.py:853-872 */
void mpm_set_velocity_on_cuboid(MPM *s, const double *point, const double *size, const double *vel, double t0, double t1, int reset) {
    BC *b = new_bc(s);
    b->type = BC_CUBOID;
    for (int d = 0; d < 3; ++d) { b->point[d] = P_(point[d]); b->size[d] = P_(size[d]); b->velocity[d] = P_(vel[d]); }
    b->start_time = P_(t0); b->end_time = P_(t1); b->reset = reset;
}
/* add_bounding_box, mpm_solver_warp.py:910-915 */
void mpm_add_bounding_box(MPM *s, double t0, double t1) {
    BC *b = new_bc(s);
    b->type = BC_BBOX; b->start_time = P_(t0); b->end_time = P_(t1);
}

static PMod *new_pmod(MPM *s) {
    s->pmods = (PMod *)realloc(s->pmods, (size_t)(s->n_pmod + 1) * sizeof(PMod));
    PMod *m = &s->pmods[s->n_pmod++];
    memset(m, 0, sizeof *m);
    m->mask = (int *)calloc((size_t)s->n, sizeof(int));
    return m;
}
/* box selection: selection_add_impulse_on_particles / selection_enforce_particle_velocity_translation,
 * mpm_utils.py:613-642 */
static void select_box(MPM *s, PMod *m) {
    for (int p = 0; p < s->n; ++p) {
        real o[3] = {s->x[3 * p] - m->point[0], s->x[3 * p + 1] - m->point[1], s->x[3 * p + 2] - m->point[2]};
        m->mask[p] = (r_abs(o[0]) < m->size[0] && r_abs(o[1]) < m->size[1] && r_abs(o[2]) < m->size[2]) ? 1 : 0;
    }
}
/* add_impulse_on_particles, mpm_solver_warp.py:982-1013 */
void mpm_add_impulse(MPM *s, const double *force, double dt, const double *point, const double *size, int num_dt, double t0) {
    PMod *m = new_pmod(s);
    m->type = PM_IMPULSE;
    m->start_time = P_(t0); m->end_time = P_((t0 + dt * num_dt));
    for (int d = 0; d < 3; ++d) { m->point[d] = P_(point[d]); m->size[d] = P_(size[d]); m->force[d] = P_(force[d]); }
    select_box(s, m);
}
/* enforce_particle_velocity_translation, mpm_solver_warp.py:1031-1059 */
void mpm_enforce_translation(MPM *s, const double *point, const double *size, const double *vel, double t0, double t1) {
    PMod *m = new_pmod(s);
    m->type = PM_TRANSLATION;
    m->start_time = P_(t0); m->end_time = P_(t1);
    for (int d = 0; d < 3; ++d) { m->point[d] = P_(point[d]); m->size[d] = P_(size[d]); m->velocity[d] = P_(vel[d]); }
    select_box(s, m);
}
/* enforce_particle_velocity_rotation, mpm_solver_warp.py:1080-1135 + selection kernel mpm_utils.py:645-663.
 * The horizontal axes are prepared by the caller exactly as :1105-1117 does (host-side float maths). */
void mpm_enforce_rotation(MPM *s, const double *point, const double *normal, const double *h1, const double *h2,
                          double half_height, double radius, double rotation_scale, double translation_scale, double t0, double t1) {
    PMod *m = new_pmod(s);
    m->type = PM_ROTATION;
    for (int d = 0; d < 3; ++d) { m->point[d] = P_(point[d]); m->normal[d] = P_(normal[d]); m->h1[d] = P_(h1[d]); m->h2[d] = P_(h2[d]); }
    m->half_height = P_(half_height); m->radius = P_(radius);
    m->rotation_scale = P_(rotation_scale); m->translation_scale = P_(translation_scale);
    m->start_time = P_(t0); m->end_time = P_(t1);
    for (int p = 0; p < s->n; ++p) {
        real o[3] = {s->x[3 * p] - m->point[0], s->x[3 * p + 1] - m->point[1], s->x[3 * p + 2] - m->point[2]};
        real dn = o[0] * m->normal[0] + o[1] * m->normal[1] + o[2] * m->normal[2];
        real vert = r_abs(dn);
        real h[3] = {o[0] - dn * m->normal[0], o[1] - dn * m->normal[1], o[2] - dn * m->normal[2]};
        m->mask[p] = (vert < m->half_height && v_len(h) < m->radius) ? 1 : 0;
    }
}

/* ------------------------------------------------------------------ stress models */
/* kirchoff_stress_FCR, mpm_utils.py:10-17 */
static void stress_FCR(const real *F, const real *U, const real *V, real J, real mu, real lam, real *out) {
    real Vt[9], Rm[9], Ft[9], D[9], T[9];
    m_T(V, Vt); m_mul(U, Vt, Rm); m_T(F, Ft);
    for (int i = 0; i < 9; ++i) D[i] = R_(2.0) * mu * (F[i] - Rm[i]);
    m_mul(D, Ft, T);
    real iso = lam * J * (J - R_(1.0));
    for (int i = 0; i < 9; ++i) out[i] = T[i];
    out[0] += iso; out[4] += iso; out[8] += iso;
}
/* kirchoff_stress_water, mpm_utils.py:20-28 */
static void stress_water(real J, real bulk, real *out) {
    real gamma = R_(1.1);
    real pressure = -bulk * (r_pow(J, -gamma) - R_(1.));
    m_diag(J * pressure, J * pressure, J * pressure, out);
}
/* kirchoff_stress_StVK, mpm_utils.py:52-68 */
static void stress_StVK(const real *F, const real *U, const real *V, const real *sig_in, real mu, real lam, real *out) {
    real sig[3] = {r_max(sig_in[0], R_(0.01)), r_max(sig_in[1], R_(0.01)), r_max(sig_in[2], R_(0.01))};
    real eps[3] = {r_log(sig[0]), r_log(sig[1]), r_log(sig[2])};
    real lss = r_log(sig[0]) + r_log(sig[1]) + r_log(sig[2]);
    real tau[3];
    for (int d = 0; d < 3; ++d) tau[d] = R_(2.0) * mu * eps[d] + lam * lss * R_(1.0);
    real D[9], Vt[9], Ft[9], T[9];
    m_diag(tau[0], tau[1], tau[2], D);
    m_T(V, Vt); m_T(F, Ft);
    m_mul(U, D, T); m_mul(T, Vt, T); m_mul(T, Ft, out);
}
/* kirchoff_stress_drucker_prager, mpm_utils.py:71-86 */
static void stress_DP(const real *F, const real *U, const real *V, const real *sig, real mu, real lam, real *out) {
    real lss = r_log(sig[0]) + r_log(sig[1]) + r_log(sig[2]);
    real c[3];
    for (int d = 0; d < 3; ++d)
        c[d] = R_(2.0) * mu * r_log(sig[d]) * (R_(1.0) / sig[d]) + lam * lss * (R_(1.0) / sig[d]);
    real D[9], Vt[9], Ft[9], T[9];
    m_diag(c[0], c[1], c[2], D);
    m_T(V, Vt); m_T(F, Ft);
    m_mul(U, D, T); m_mul(T, Vt, T); m_mul(T, Ft, out);
}

/* ------------------------------------------------------------------ return mappings */
/* von_mises_return_mapping (damage=0), mpm_utils.py:89-135;
 * von_mises_return_mapping_with_damage (damage=1), mpm_utils.py:138-191 */
static void rm_von_mises(MPM *s, int p, const real *Ft, real *Fout, int damage) {
    real U[9], V[9], so[3];
    svd3(Ft, U, so, V);
    real sig[3] = {r_max(so[0], R_(0.01)), r_max(so[1], R_(0.01)), r_max(so[2], R_(0.01))};
    real eps[3] = {r_log(sig[0]), r_log(sig[1]), r_log(sig[2])};
    real temp = (eps[0] + eps[1] + eps[2]) / R_(3.0);
    real tau[3], sum_tau = 0;
    for (int d = 0; d < 3; ++d) tau[d] = R_(2.0) * s->mu[p] * eps[d] + s->lam[p] * (eps[0] + eps[1] + eps[2]) * R_(1.0);
    sum_tau = tau[0] + tau[1] + tau[2];
    real cond[3] = {tau[0] - sum_tau / R_(3.0), tau[1] - sum_tau / R_(3.0), tau[2] - sum_tau / R_(3.0)};
    if (v_len(cond) > s->yield_stress[p]) {
        if (damage && s->yield_stress[p] <= 0) { memcpy(Fout, Ft, 9 * sizeof(real)); return; }
        real eh[3] = {eps[0] - temp, eps[1] - temp, eps[2] - temp};
        real ehn = v_len(eh) + R_(1e-6);
        real dg = ehn - s->yield_stress[p] / (R_(2.0) * s->mu[p]);
        for (int d = 0; d < 3; ++d) eps[d] = eps[d] - (dg / ehn) * eh[d];
        if (damage) {
            real sc[3] = {(dg / ehn) * eh[0], (dg / ehn) * eh[1], (dg / ehn) * eh[2]};
            s->yield_stress[p] = s->yield_stress[p] - s->softening * v_len(sc);
            if (s->yield_stress[p] <= 0) { s->mu[p] = 0; s->lam[p] = 0; }
        }
        real D[9], Vt[9], T[9];
        m_diag(r_exp(eps[0]), r_exp(eps[1]), r_exp(eps[2]), D);
        m_T(V, Vt); m_mul(U, D, T); m_mul(T, Vt, Fout);
        if (s->hardening == R_(1.0))
            s->yield_stress[p] = s->yield_stress[p] + R_(2.0) * s->mu[p] * s->xi * dg;
    } else {
        memcpy(Fout, Ft, 9 * sizeof(real));
    }
}
/* viscoplasticity_return_mapping_with_StVK, mpm_utils.py:195-239 */
static void rm_visco(MPM *s, int p, const real *Ft, real dt, real *Fout) {
    real U[9], V[9], so[3];
    svd3(Ft, U, so, V);
    real sig[3] = {r_max(so[0], R_(0.01)), r_max(so[1], R_(0.01)), r_max(so[2], R_(0.01))};
    real b[3] = {sig[0] * sig[0], sig[1] * sig[1], sig[2] * sig[2]};
    real eps[3] = {r_log(sig[0]), r_log(sig[1]), r_log(sig[2])};
    real tr = eps[0] + eps[1] + eps[2];
    real eh[3] = {eps[0] - tr / R_(3.0), eps[1] - tr / R_(3.0), eps[2] - tr / R_(3.0)};
    real st[3] = {R_(2.0) * s->mu[p] * eh[0], R_(2.0) * s->mu[p] * eh[1], R_(2.0) * s->mu[p] * eh[2]};
    real stn = v_len(st);
    real y = stn - r_sqrt(R_(2.0) / R_(3.0)) * s->yield_stress[p];
    if (y > 0) {
        real mu_hat = s->mu[p] * (b[0] + b[1] + b[2]) / R_(3.0);
        real snn = stn - y / (R_(1.0) + s->plastic_viscosity / (R_(2.0) * mu_hat * dt));
        real en[3];
        for (int d = 0; d < 3; ++d) en[d] = R_(1.0) / (R_(2.0) * s->mu[p]) * ((snn / stn) * st[d]) + tr / R_(3.0);
        real D[9], Vt[9], T[9];
        m_diag(r_exp(en[0]), r_exp(en[1]), r_exp(en[2]), D);
        m_T(V, Vt); m_mul(U, D, T); m_mul(T, Vt, Fout);
    } else {
        memcpy(Fout, Ft, 9 * sizeof(real));
    }
}
/* sand_return_mapping, mpm_utils.py:242-279 */
static void rm_sand(MPM *s, int p, const real *Ft, real *Fout) {
    real U[9], V[9], sig[3];
    svd3(Ft, U, sig, V);
    real eps[3] = {r_log(r_max(r_abs(sig[0]), R_(1e-14))), r_log(r_max(r_abs(sig[1]), R_(1e-14))), r_log(r_max(r_abs(sig[2]), R_(1e-14)))};
    real tr = eps[0] + eps[1] + eps[2];
    real eh[3] = {eps[0] - tr / R_(3.0), eps[1] - tr / R_(3.0), eps[2] - tr / R_(3.0)};
    real ehn = v_len(eh);
    real dg = ehn + (R_(3.0) * s->lam[p] + R_(2.0) * s->mu[p]) / (R_(2.0) * s->mu[p]) * tr * s->alpha;
    if (dg <= 0) memcpy(Fout, Ft, 9 * sizeof(real));
    if (dg > 0 && tr > 0) { real Vt[9]; m_T(V, Vt); m_mul(U, Vt, Fout); }
    if (dg > 0 && tr <= 0) {
        real H[3];
        for (int d = 0; d < 3; ++d) H[d] = eps[d] - eh[d] * (dg / ehn);
        real D[9], Vt[9], T[9];
        m_diag(r_exp(H[0]), r_exp(H[1]), r_exp(H[2]), D);
        m_T(V, Vt); m_mul(U, D, T); m_mul(T, Vt, Fout);
    }
}

/* ------------------------------------------------------------------ kernels */
/* zero_grid, mpm_utils.py:295-300 */
/* Activity windows.  In the reference `time`, `start_time` and `end_time` reach the kernels as 32-bit floats whatever
 * the host computes in, so the DECISION "is this modifier / collider active in this substep" is a float32 comparison.
 * The float64 build keeps that decision in float32 (it is discrete: with time accumulated in double, 30 x 1e-4 is
 * 0.00299999999999999 < 3e-3 whereas both round to the same float32) and only does the arithmetic in double. */
#define WIN(t, a, b) ((float)(t) >= (float)(a) && (float)(t) < (float)(b))

void mpm_zero_grid(MPM *s) {
    size_t G = (size_t)s->ng * s->ng * s->ng;
    memset(s->grid_m, 0, G * sizeof(real));
    memset(s->grid_v_in, 0, 3 * G * sizeof(real));
    memset(s->grid_v_out, 0, 3 * G * sizeof(real));
}

/* pre-P2G particle modifiers: apply_force mpm_solver_warp.py:1015-1027;
 * modify_particle_v_before_p2g (translation) :1061-1073; (rotation) :1137-1179 */
void mpm_pre_p2g(MPM *s, double dt_d) {
    real time = P_(s->time), dt = P_(dt_d);
    for (int k = 0; k < s->n_pmod; ++k) { /* impulses first (:529-535), in registration order */
        PMod *m = &s->pmods[k];
        if (m->type != PM_IMPULSE) continue;
        if (WIN(time, m->start_time, m->end_time))
            for (int p = 0; p < s->n; ++p)
                if (m->mask[p] == 1) {
                    real imp[3] = {m->force[0] / s->mass[p], m->force[1] / s->mass[p], m->force[2] / s->mass[p]};
                    for (int d = 0; d < 3; ++d) s->v[3 * p + d] = s->v[3 * p + d] + imp[d] * dt;
                }
    }
    for (int k = 0; k < s->n_pmod; ++k) { /* then velocity modifiers (:537-547) */
        PMod *m = &s->pmods[k];
        if (m->type == PM_IMPULSE) continue;
        if (!WIN(time, m->start_time, m->end_time)) continue;
        for (int p = 0; p < s->n; ++p) {
            if (m->mask[p] != 1) continue;
            if (m->type == PM_TRANSLATION) {
                for (int d = 0; d < 3; ++d) s->v[3 * p + d] = m->velocity[d];
            } else {
                real o[3] = {s->x[3 * p] - m->point[0], s->x[3 * p + 1] - m->point[1], s->x[3 * p + 2] - m->point[2]};
                real dn = o[0] * m->normal[0] + o[1] * m->normal[1] + o[2] * m->normal[2];
                real h[3] = {o[0] - dn * m->normal[0], o[1] - dn * m->normal[1], o[2] - dn * m->normal[2]};
                real hd = v_len(h);
                real cosine = (o[0] * m->h1[0] + o[1] * m->h1[1] + o[2] * m->h1[2]) / hd;
                real theta = sizeof(real) == 4 ? (real)acosf((float)cosine) : (real)acos((double)cosine);
                if (!((o[0] * m->h2[0] + o[1] * m->h2[1] + o[2] * m->h2[2]) > 0)) theta = -theta;
                real sn = sizeof(real) == 4 ? (real)sinf((float)theta) : (real)sin((double)theta);
                real cs = sizeof(real) == 4 ? (real)cosf((float)theta) : (real)cos((double)theta);
                real a1 = -hd * sn * m->rotation_scale, a2 = hd * cs * m->rotation_scale, av = m->translation_scale;
                for (int d = 0; d < 3; ++d) s->v[3 * p + d] = a1 * m->h1[d] + a2 * m->h2[d] + av * m->normal[d];
            }
        }
    }
}

/* compute_stress_from_F_trial, mpm_utils.py:467-526 */
void mpm_compute_stress(MPM *s, double dt_d) {
    real dt = P_(dt_d);
    OMP_FOR
    for (int p = 0; p < s->n; ++p) {
        int material = s->material[p];
        if (s->selection[p] != 0) continue;
        real *F = s->F + 9 * p;
        const real *Ft = s->F_trial + 9 * p;
        if (material == 1) rm_von_mises(s, p, Ft, F, 0);
        else if (material == 2) rm_sand(s, p, Ft, F);
        else if (material == 3) rm_visco(s, p, Ft, dt, F);
        else if (material == 5) rm_von_mises(s, p, Ft, F, 1);
        else memcpy(F, Ft, 9 * sizeof(real));
#ifdef ORACLE_EXPERIMENT
        if ((ORACLE_EXPERIMENT & 2) && s->exp_init) {   /* F = returnMap(F_trial): the double copy follows (exactly, where F = F_trial) */
            int same = 1;
            for (int i = 0; i < 9; ++i) same = same && (F[i] == Ft[i]);
            for (int i = 0; i < 9; ++i) s->Fd[9 * p + i] = same ? s->Ftd[9 * p + i] : (double)F[i];
        }
#endif
        real J = m_det(F);
        real U[9], V[9], sig[3], st[9];
        memset(st, 0, sizeof st);
        svd3(F, U, sig, V);
        if (material == 0 || material == 5) stress_FCR(F, U, V, J, s->mu[p], s->lam[p], st);
        if (material == 1) stress_StVK(F, U, V, sig, s->mu[p], s->lam[p], st);
        if (material == 2) stress_DP(F, U, V, sig, s->mu[p], s->lam[p], st);
        if (material == 3) stress_StVK(F, U, V, sig, s->mu[p], s->lam[p], st);
        if (material == 6) stress_water(J, s->bulk[p], st);
        real stT[9];
        m_T(st, stT);
        for (int i = 0; i < 9; ++i) s->stress[9 * p + i] = (st[i] + stT[i]) / R_(2.0);
    }
}

/* shared stencil set-up of p2g/g2p: mpm_utils.py:343-358 / :418-434.
 * w[d][i] = weight of offset i along axis d (wp.mat33(v0,v1,v2) stacks the vectors as COLUMNS). */
static void stencil(const MPM *s, const real *x, int *base, real *fx, real w[3][3], real dw[3][3]) {
    for (int d = 0; d < 3; ++d) {
        real gp = x[d] * s->inv_dx;
        base[d] = (int)(gp - R_(0.5)); /* wp.int truncates toward zero */
        fx[d] = gp - (real)base[d];
        real wa = R_(1.5) - fx[d], wb = fx[d] - R_(1.0), wc = fx[d] - R_(0.5);
        w[d][0] = wa * wa * R_(0.5);
        w[d][1] = R_(0.0) - wb * wb + R_(0.75);
        w[d][2] = wc * wc * R_(0.5);
        dw[d][0] = fx[d] - R_(1.5);
        dw[d][1] = R_(-2.0) * (fx[d] - R_(1.0));
        dw[d][2] = fx[d] - R_(0.5);
    }
}
static int stencil_inside(const MPM *s, const int *base) {
    for (int d = 0; d < 3; ++d)
        if (base[d] < 0 || base[d] + 2 >= s->ng) return 0;
    return 1;
}

/* p2g_apic_with_stress, mpm_utils.py:338-394: one particle.  `atomic`: the adds are `omp atomic` (threads may share nodes) */
static inline void p2g_particle(MPM *s, int p, real dt, int atomic) {
    if (s->selection[p] != 0) return;
    const real *stress = s->stress + 9 * p;
    int base[3]; real fx[3], w[3][3], dw[3][3];
    stencil(s, s->x + 3 * p, base, fx, w, dw);
    if (!stencil_inside(s, base)) { /* reference: no bounds check (UB) */
        OMP_ATOMIC
        s->oob++;
        return;
    }
    /* C' and -vol*stress do not depend on (i,j,k); the reference recomputes them per node
     * (:372-381) with identical operands, so hoisting is value-preserving. */
    real C[9], nvs[9];
    const real *Cp = s->C + 9 * p;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            C[3 * a + b] = (R_(1.0) - s->rpic_damping) * Cp[3 * a + b] +
                           s->rpic_damping / R_(2.0) * (Cp[3 * a + b] - Cp[3 * b + a]);
    if (s->rpic_damping < R_(-0.001)) memset(C, 0, sizeof C);
    for (int a = 0; a < 9; ++a) nvs[a] = -s->vol[p] * stress[a];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                real dpos[3] = {((real)i - fx[0]) * s->dx, ((real)j - fx[1]) * s->dx, ((real)k - fx[2]) * s->dx};
                real weight = w[0][i] * w[1][j] * w[2][k];
                real dweight[3] = {dw[0][i] * w[1][j] * w[2][k] * s->inv_dx, w[0][i] * dw[1][j] * w[2][k] * s->inv_dx,
                                   w[0][i] * w[1][j] * dw[2][k] * s->inv_dx}; /* compute_dweight :303-312 */
                real ef[3], Cd[3];
                m_vec(nvs, dweight, ef);
                m_vec(C, dpos, Cd);
                real wm = weight * s->mass[p];
                size_t gi = GI(s, base[0] + i, base[1] + j, base[2] + k);
                if (atomic) {
                    for (int d = 0; d < 3; ++d) {
                        real add = wm * (s->v[3 * p + d] + Cd[d]) + dt * ef[d];
                        OMP_ATOMIC
                        s->grid_v_in[3 * gi + d] += add;          /* wp.atomic_add, mpm_utils.py:393 */
                    }
                    OMP_ATOMIC
                    s->grid_m[gi] += wm;                           /* :394 */
                } else {
                    for (int d = 0; d < 3; ++d) s->grid_v_in[3 * gi + d] += wm * (s->v[3 * p + d] + Cd[d]) + dt * ef[d];
                    s->grid_m[gi] += wm;
                }
            }
}

#ifdef _OPENMP
/* The multi-core build (bench.py's cpu_baseline only).  Atomic float adds from 256 threads into one grid scale to ~2 cores
 * (every add moves a cache line between cores), so this build scatters WITHOUT atomics: the particles are counting-sorted into
 * 4^3-cell tiles (the base cell of the stencil decides), the tiles are coloured by the parity of their three indices, and the
 * eight colours run one after the other -- two tiles of one colour are at least 8 cells apart on some axis while a stencil
 * reaches 2 cells beyond its base, so the threads of a colour never touch the same node.  Same arithmetic per particle; the
 * order of the sums differs from the serial builds (which the tests use). */
static int *g_tile_of = NULL, *g_tile_start = NULL, *g_tile_fill = NULL, *g_sorted = NULL;
static int g_cap_n = 0, g_cap_t = 0;
void mpm_p2g(MPM *s, double dt_d) {
    real dt = P_(dt_d);
    const int nt = (s->ng + 3) / 4, T = nt * nt * nt, n = s->n;
    if (n > g_cap_n) { free(g_tile_of); free(g_sorted); g_tile_of = malloc(sizeof(int) * n); g_sorted = malloc(sizeof(int) * n); g_cap_n = n; }
    if (T + 1 > g_cap_t) { free(g_tile_start); free(g_tile_fill); g_tile_start = malloc(sizeof(int) * (T + 1)); g_tile_fill = malloc(sizeof(int) * (T + 1)); g_cap_t = T + 1; }
    OMP_FOR
    for (int p = 0; p < n; ++p) {
        int base[3]; real fx[3], w[3][3], dw[3][3];
        stencil(s, s->x + 3 * p, base, fx, w, dw);
        int t[3];
        for (int d = 0; d < 3; ++d) { int b = base[d] < 0 ? 0 : (base[d] >= s->ng ? s->ng - 1 : base[d]); t[d] = b / 4; }
        g_tile_of[p] = (t[0] * nt + t[1]) * nt + t[2];
    }
    memset(g_tile_start, 0, sizeof(int) * (T + 1));
    for (int p = 0; p < n; ++p) g_tile_start[g_tile_of[p] + 1]++;
    for (int t = 0; t < T; ++t) g_tile_start[t + 1] += g_tile_start[t];
    memcpy(g_tile_fill, g_tile_start, sizeof(int) * (T + 1));
    for (int p = 0; p < n; ++p) g_sorted[g_tile_fill[g_tile_of[p]]++] = p;
    for (int colour = 0; colour < 8; ++colour) {
        #pragma omp parallel for schedule(dynamic, 1)
        for (int t = 0; t < T; ++t) {
            const int tz = t % nt, ty = (t / nt) % nt, tx = t / (nt * nt);
            if ((((tx & 1) << 2) | ((ty & 1) << 1) | (tz & 1)) != colour) continue;
            for (int q = g_tile_start[t]; q < g_tile_start[t + 1]; ++q) p2g_particle(s, g_sorted[q], dt, 0);
        }
    }
}
#else
void mpm_p2g(MPM *s, double dt_d) {
    real dt = P_(dt_d);
    for (int p = 0; p < s->n; ++p) p2g_particle(s, p, dt, 1);
}
#endif

/* grid_normalization_and_gravity, mpm_utils.py:398-409 */
void mpm_grid_update(MPM *s, double dt_d) {
    real dt = P_(dt_d);
    size_t G = (size_t)s->ng * s->ng * s->ng;
    OMP_FOR
    for (size_t gi = 0; gi < G; ++gi)
        if (s->grid_m[gi] > R_(1e-15)) {
            real inv = R_(1.0) / s->grid_m[gi];
            for (int d = 0; d < 3; ++d) s->grid_v_out[3 * gi + d] = s->grid_v_in[3 * gi + d] * inv + dt * s->g[d];
        }
}
/* add_damping_via_grid, mpm_utils.py:583-588 (gate mpm_solver_warp.py:595) */
void mpm_grid_damping(MPM *s) {
    if (!(s->grid_v_damping_scale < R_(1.0))) return;
    size_t G = (size_t)s->ng * s->ng * s->ng;
    OMP_FOR
    for (size_t i = 0; i < 3 * G; ++i) s->grid_v_out[i] = s->grid_v_out[i] * s->grid_v_damping_scale;
}

/* grid BC kernels + host modify: surface mpm_solver_warp.py:785-840; cuboid :874-905; bounding box :917-974 */
void mpm_apply_bcs(MPM *s, double dt_d) {
    real time = P_(s->time), dt = P_(dt_d);
    int ng = s->ng;
    for (int k = 0; k < s->n_bc; ++k) {
        BC *b = &s->bcs[k];
        OMP_FOR
        for (int ix = 0; ix < ng; ++ix)
            for (int iy = 0; iy < ng; ++iy)
                for (int iz = 0; iz < ng; ++iz) {
                    real *vo = s->grid_v_out + 3 * GI(s, ix, iy, iz);
                    if (b->type == BC_SURFACE) {
                        if (WIN(time, b->start_time, b->end_time)) {
                            real off[3] = {(real)ix * s->dx - b->point[0], (real)iy * s->dx - b->point[1], (real)iz * s->dx - b->point[2]};
                            real dp = off[0] * b->normal[0] + off[1] * b->normal[1] + off[2] * b->normal[2];
                            if (dp < R_(0.0)) {
                                if (b->surface_type == 0) {
                                    vo[0] = vo[1] = vo[2] = 0;
                                } else if (b->surface_type == 11) {
                                    if ((real)iz * s->dx < R_(0.4) || (real)iz * s->dx > R_(0.53)) {
                                        vo[0] = vo[1] = vo[2] = 0;
                                    } else {
                                        real v0 = vo[0], v2 = vo[2];
                                        vo[0] = v0 * R_(0.3); vo[1] = R_(0.0) * R_(0.3); vo[2] = v2 * R_(0.3);
                                    }
                                } else {
                                    /* :821-840: slip/friction maths is dead, the store is zero */
                                    vo[0] = vo[1] = vo[2] = 0;
                                }
                            }
                        }
                    } else if (b->type == BC_CUBOID) {
                        if (WIN(time, b->start_time, b->end_time)) {
                            real off[3] = {(real)ix * s->dx - b->point[0], (real)iy * s->dx - b->point[1], (real)iz * s->dx - b->point[2]};
                            if (r_abs(off[0]) < b->size[0] && r_abs(off[1]) < b->size[1] && r_abs(off[2]) < b->size[2]) {
                                vo[0] = b->velocity[0]; vo[1] = b->velocity[1]; vo[2] = b->velocity[2];
                            }
                        } else if (b->reset == 1) {
                            if ((float)time < (float)b->end_time + 15.0f * (float)dt) { vo[0] = vo[1] = vo[2] = 0; }
                        }
                    } else if (b->type == BC_BBOX) {
                        int padding = 3;
                        if (WIN(time, b->start_time, b->end_time)) {
                            if (ix < padding && vo[0] < 0) vo[0] = 0;
                            if (ix >= ng - padding && vo[0] > 0) vo[0] = 0;
                            if (iy < padding && vo[1] < 0) vo[1] = 0;
                            if (iy >= ng - padding && vo[1] > 0) vo[1] = 0;
                            if (iz < padding && vo[2] < 0) vo[2] = 0;
                            if (iz >= ng - padding && vo[2] > 0) vo[2] = 0;
                        }
                    }
                }
        if (b->type == BC_CUBOID) { /* host `modify`, :899-905: python-float arithmetic, stored back as f32 vec3 */
            if (s->time >= (double)(float)b->start_time && s->time < (double)(float)b->end_time)   /* the struct members are float32 */
                for (int d = 0; d < 3; ++d) b->point[d] = P_((double)b->point[d] + dt_d * (double)b->velocity[d]);
        }
    }
}

/* g2p, mpm_utils.py:412-463 (update_cov_with_F is always False in the reference flows) */
void mpm_g2p(MPM *s, double dt_d) {
    real dt = P_(dt_d);
#ifdef ORACLE_EXPERIMENT
    if (!s->exp_init) {
        const size_t N = (size_t)s->n;
        s->xd = malloc(3 * N * sizeof(double)); s->Fd = malloc(9 * N * sizeof(double)); s->Ftd = malloc(9 * N * sizeof(double));
        for (size_t i = 0; i < 3 * N; ++i) s->xd[i] = (double)s->x[i];
        for (size_t i = 0; i < 9 * N; ++i) { s->Fd[i] = (double)s->F[i]; s->Ftd[i] = (double)s->F_trial[i]; }
        s->exp_init = 1;
    }
#endif
    OMP_FOR
    for (int p = 0; p < s->n; ++p) {
        if (s->selection[p] != 0) continue;
        int base[3]; real fx[3], w[3][3], dw[3][3];
        stencil(s, s->x + 3 * p, base, fx, w, dw);
        if (!stencil_inside(s, base)) {
            OMP_ATOMIC
            s->oob++;
            continue;
        }
        real nv[3] = {0, 0, 0}, nC[9] = {0}, nF[9] = {0};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) {
                    real dpos[3] = {(real)i - fx[0], (real)j - fx[1], (real)k - fx[2]};
                    real weight = w[0][i] * w[1][j] * w[2][k];
                    const real *gv = s->grid_v_out + 3 * GI(s, base[0] + i, base[1] + j, base[2] + k);
                    real dweight[3] = {dw[0][i] * w[1][j] * w[2][k] * s->inv_dx, w[0][i] * dw[1][j] * w[2][k] * s->inv_dx,
                                       w[0][i] * w[1][j] * dw[2][k] * s->inv_dx};
                    real sc = weight * s->inv_dx * R_(4.0);
                    for (int a = 0; a < 3; ++a) {
                        nv[a] = nv[a] + gv[a] * weight;
                        for (int b = 0; b < 3; ++b) {
                            nC[3 * a + b] = nC[3 * a + b] + (gv[a] * dpos[b]) * sc;
                            nF[3 * a + b] = nF[3 * a + b] + gv[a] * dweight[b];
                        }
                    }
                }
        for (int d = 0; d < 3; ++d) {
            s->v[3 * p + d] = nv[d];
            s->x[3 * p + d] = s->x[3 * p + d] + dt * nv[d];
        }
        memcpy(s->C + 9 * p, nC, sizeof nC);
        real A[9];
        for (int i = 0; i < 9; ++i) A[i] = ((i % 4 == 0) ? R_(1.0) : R_(0.0)) + nF[i] * dt;
        m_mul(A, s->F + 9 * p, s->F_trial + 9 * p);
#ifdef ORACLE_EXPERIMENT
        if (ORACLE_EXPERIMENT & 1)      /* x = the float rounding of a position accumulated in double (the same float32 v) */
            for (int d = 0; d < 3; ++d) { s->xd[3 * p + d] += (double)dt * (double)nv[d]; s->x[3 * p + d] = (real)s->xd[3 * p + d]; }
        if (ORACLE_EXPERIMENT & 2) {    /* F_trial = the float rounding of (I + dt grad v) F accumulated in double (the same float32 grad v) */
            double Ad[9], Td[9];
            for (int i = 0; i < 9; ++i) Ad[i] = ((i % 4 == 0) ? 1.0 : 0.0) + (double)nF[i] * (double)dt;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) Td[3 * i + j] = Ad[3 * i] * s->Fd[9 * p + j] + Ad[3 * i + 1] * s->Fd[9 * p + 3 + j] + Ad[3 * i + 2] * s->Fd[9 * p + 6 + j];
            for (int i = 0; i < 9; ++i) { s->Ftd[9 * p + i] = Td[i]; s->F_trial[9 * p + i] = (real)Td[i]; }
        }
#endif
    }
}

/* MPM_Simulator_WARP.p2g2p, mpm_solver_warp.py:514-637 */
void mpm_p2g2p(MPM *s, double dt) {
    mpm_zero_grid(s);
    mpm_pre_p2g(s, dt);
    mpm_compute_stress(s, dt);
    mpm_p2g(s, dt);
    mpm_grid_update(s, dt);
    mpm_grid_damping(s);
    mpm_apply_bcs(s, dt);
    mpm_g2p(s, dt);
    s->time = s->time + dt;
}
void mpm_run(MPM *s, double dt, int n_substeps) {
    for (int i = 0; i < n_substeps; ++i) mpm_p2g2p(s, dt);
}

/* compute_cov_from_F, mpm_utils.py:529-553: cov = F_trial * init_cov * F_trial^T (6-float upper triangles) */
void mpm_compute_cov(MPM *s, const real *init_cov, real *cov) {
    for (int p = 0; p < s->n; ++p) {
        const real *c = init_cov + 6 * p;
        real M[9] = {c[0], c[1], c[2], c[1], c[3], c[4], c[2], c[4], c[5]};
        real Ft[9], T[9];
        m_T(s->F_trial + 9 * p, Ft);
        m_mul(s->F_trial + 9 * p, M, T); m_mul(T, Ft, T);
        real *o = cov + 6 * p;
        o[0] = T[0]; o[1] = T[1]; o[2] = T[2]; o[3] = T[4]; o[4] = T[5]; o[5] = T[8];
    }
}
/* compute_R_from_F, mpm_utils.py:556-580: R^T of the polar rotation of F_trial */
void mpm_compute_R(MPM *s, real *Rout) {
    for (int p = 0; p < s->n; ++p) {
        real U[9], V[9], sig[3], Vt[9], Rm[9];
        svd3(s->F_trial + 9 * p, U, sig, V);
        if (m_det(U) < 0) { U[2] = -U[2]; U[5] = -U[5]; U[8] = -U[8]; }
        if (m_det(V) < 0) { V[2] = -V[2]; V[5] = -V[5]; V[8] = -V[8]; }
        m_T(V, Vt); m_mul(U, Vt, Rm); m_T(Rm, Rout + 9 * p);
    }
}
int mpm_sizeof_real(void) { return (int)sizeof(real); }
