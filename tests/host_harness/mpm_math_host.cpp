// tests/host_harness/mpm_math_host.cpp -- compiles pixie_amd/csrc/mpm_math.h for the HOST so
// that CPU-only tests can compare the device arithmetic against oracle/ before any GPU run.
// Test infrastructure only: the product never executes this.
#include "../../pixie_amd/csrc/mpm_math.h"
using namespace pixie;
extern "C" {
void hh_svd3(int n, const float* A, float* U, float* S, float* V) {
    for (int p = 0; p < n; ++p) {
        Mat3 F, u, v; float s[3];
        for (int i = 0; i < 9; ++i) F.m[i] = A[9 * p + i];
        svd3(F, u, s, v);
        for (int i = 0; i < 9; ++i) { U[9 * p + i] = u.m[i]; V[9 * p + i] = v.m[i]; }
        for (int i = 0; i < 3; ++i) S[3 * p + i] = s[i];
    }
}
void hh_left_stretch(int n, const float* A, float* U, float* S, int* sweeps) {
    for (int p = 0; p < n; ++p) {
        Mat3 F, u; float s[3];
        for (int i = 0; i < 9; ++i) F.m[i] = A[9 * p + i];
        left_stretch(F, mat_det(F), u, s, &sweeps[p]);
        for (int i = 0; i < 9; ++i) U[9 * p + i] = u.m[i];
        for (int i = 0; i < 3; ++i) S[3 * p + i] = s[i];
    }
}
void hh_stress(int n, const int* material, const float* Ft, float* mu, float* lam, const float* bulk, float* ys,
               float alpha, float hardening, float xi, float softening, float plastic_viscosity, float dt,
               float* Fout, float* tau) {
    MaterialScalars ms{alpha, hardening, xi, softening, plastic_viscosity};
    for (int p = 0; p < n; ++p) {
        Mat3 ft, F, T;
        for (int i = 0; i < 9; ++i) ft.m[i] = Ft[9 * p + i];
        return_map_and_stress(material[p], ft, mu[p], lam[p], bulk[p], ys[p], ms, dt, F, T);
        for (int i = 0; i < 9; ++i) { Fout[9 * p + i] = F.m[i]; tau[9 * p + i] = T.m[i]; }
    }
}
void hh_polar(int n, const float* A, float* Rout, int* ok) {
    for (int p = 0; p < n; ++p) {
        Mat3 F, R;
        for (int i = 0; i < 9; ++i) F.m[i] = A[9 * p + i];
        ok[p] = polar_rotation(F, R) ? 1 : 0;
        for (int i = 0; i < 9; ++i) Rout[9 * p + i] = R.m[i];
    }
}
void hh_stencil(int n, const float* x, float inv_dx, int* base, float* w, float* dw) {
    for (int p = 0; p < n; ++p) {
        Stencil s = make_stencil(x[3 * p], x[3 * p + 1], x[3 * p + 2], inv_dx);
        for (int d = 0; d < 3; ++d) {
            base[3 * p + d] = s.base[d];
            for (int i = 0; i < 3; ++i) { w[9 * p + 3 * d + i] = s.w[d][i]; dw[9 * p + 3 * d + i] = s.dw[d][i]; }
        }
    }
}
}
