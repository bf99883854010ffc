// pixie_amd/csrc/particle_filling.hip -- the particle pre-pass of the MPM program on gfx950.
//
// Replaces the Taichi kernels of third_party/PhysGaussian/particle_filling/filling.py that gs_simulation.py:442-482 runs before
// the solver is created:
//   densify_grids (:26-92) + compute_density (:13-23)   Gaussian opacity splat of every 3DGS kernel onto a density grid
//   fill_dense_grids (:95-121)                           top cells above the density threshold up to max particles per cell
//   internal_filling (:186-244) with collision_search (:124-149) / collision_times (:152-183)
//                                                        ray-cast inside/outside test for empty cells, then fill them
//   assign_particle_to_grid + compute_particle_volume (:247-288)   particle volume = cell volume / particles in the cell
//   get_attr_from_closest (:383-403)                     nearest original Gaussian of every new particle
// All HBM-bound gather/scatter work on small arrays; nothing here is GEMM-shaped.  The new particles' positions inside
// their cell are random in the reference (ti.random()); here a counter-based hash of (seed, cell, k) makes them
// reproducible.  mcubes smoothing (smooth=True) is host-side third-party code and is not reproduced.
#include <hip/hip_runtime.h>

#include "../../include/pixie_hip.h"
#include "common.h"
#include "mpm_math.h"

namespace pixie {

struct FillGrid {
    int n;            // cells per axis
    float dx;
    int* count;       // particles per cell
    float* density;   // splatted opacity density
};

// compute_density, filling.py:13-23: mean over the cell's 8 corners of opacity * exp(-0.5 d^T Cinv d)
__device__ __forceinline__ float cell_density(int ci, int cj, int ck, const float p[3], float opacity, const Mat3& Cinv, float dx) {
    float g = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float d[3] = {p[0] - (float)(ci + i) * dx, p[1] - (float)(cj + j) * dx, p[2] - (float)(ck + k) * dx};
                float q = 0.0f;
#pragma unroll
                for (int a = 0; a < 3; ++a) q += d[a] * (Cinv.m[3 * a] * d[0] + Cinv.m[3 * a + 1] * d[1] + Cinv.m[3 * a + 2] * d[2]);
                g += expf(-0.5f * q);
            }
    return opacity * g / 8.0f;
}

// densify_grids, filling.py:26-92.  One thread per Gaussian; cov6 = (xx, xy, xz, yy, yz, zz).
__global__ __launch_bounds__(128) void densify_kernel(const float* __restrict__ pos, const float* __restrict__ opacity,
                                                      const float* __restrict__ cov6, int n, FillGrid G) {
    const int pi = blockIdx.x * 128 + threadIdx.x;
    if (pi >= n) return;
    const float p[3] = {pos[3 * pi], pos[3 * pi + 1], pos[3 * pi + 2]};
    const int ci = (int)floorf(p[0] / G.dx), cj = (int)floorf(p[1] / G.dx), ck = (int)floorf(p[2] / G.dx);
    // the reference indexes grid[i, j, k] unchecked (undefined behaviour outside the grid); such particles are not counted here
    if ((unsigned)ci < (unsigned)G.n && (unsigned)cj < (unsigned)G.n && (unsigned)ck < (unsigned)G.n)
        atomicAdd(&G.count[((size_t)ci * G.n + cj) * G.n + ck], 1);
    const float* c = cov6 + 6 * (size_t)pi;
    Mat3 S, Q;
    S.m[0] = c[0]; S.m[1] = c[1]; S.m[2] = c[2];
    S.m[3] = c[1]; S.m[4] = c[3]; S.m[5] = c[4];
    S.m[6] = c[2]; S.m[7] = c[4]; S.m[8] = c[5];
    Q = mat_identity();
    for (int sweep = 0; sweep < 8; ++sweep) {            // ti.sym_eig: cyclic Jacobi on the symmetric 3x3
        jacobi_rotate<0, 1>(S, Q);
        jacobi_rotate<0, 2>(S, Q);
        jacobi_rotate<1, 2>(S, Q);
    }
    float sig[3] = {fmaxf(S.m[0], 1e-8f), fmaxf(S.m[4], 1e-8f), fmaxf(S.m[8], 1e-8f)};
    const float inv[3] = {1.0f / sig[0], 1.0f / sig[1], 1.0f / sig[2]};
    const Mat3 Cinv = mat_udvt(Q, inv, Q);               // Q diag(1/sig) Q^T
    const float rmax = sqrtf(fmaxf(fmaxf(sig[0], sig[1]), sig[2]));
    const int r = (int)ceilf(rmax / G.dx);
    const float op = opacity[pi];
    for (int a = -r; a <= r; ++a)
        for (int b = -r; b <= r; ++b)
            for (int cz = -r; cz <= r; ++cz) {
                const int i = ci + a, j = cj + b, k = ck + cz;
                if ((unsigned)i >= (unsigned)G.n || (unsigned)j >= (unsigned)G.n || (unsigned)k >= (unsigned)G.n) continue;
                unsafeAtomicAdd(&G.density[((size_t)i * G.n + j) * G.n + k], cell_density(i, j, k, p, op, Cinv, G.dx));
            }
}

// counter-based uniform in [0, 1): 24 random bits from a 2-round integer hash of (seed, cell, sample, axis)
__device__ __forceinline__ float hash_uniform(unsigned seed, unsigned cell, unsigned k, unsigned axis) {
    unsigned x = seed ^ (cell * 0x9E3779B1u) ^ (k * 0x85EBCA77u) ^ (axis * 0xC2B2AE3Du);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ void emit_particles(int i, int j, int k, int diff, const FillGrid& G, float* __restrict__ out, long long max_samples,
                                               unsigned long long* __restrict__ counter, unsigned seed) {
    const unsigned long long start = atomicAdd(counter, (unsigned long long)diff);
    const unsigned cell = (unsigned)(((size_t)i * G.n + j) * G.n + k);
    for (int q = 0; q < diff; ++q) {
        const unsigned long long idx = start + q;
        if ((long long)idx >= max_samples) return;   // the reference writes past max_samples; here the host reports the overflow
        out[3 * idx] = ((float)i + hash_uniform(seed, cell, q, 0)) * G.dx;
        out[3 * idx + 1] = ((float)j + hash_uniform(seed, cell, q, 1)) * G.dx;
        out[3 * idx + 2] = ((float)k + hash_uniform(seed, cell, q, 2)) * G.dx;
    }
}

// fill_dense_grids, filling.py:95-121
__global__ __launch_bounds__(256) void fill_dense_kernel(FillGrid G, float thres, int max_ppc, float* __restrict__ out, long long max_samples,
                                                         unsigned long long* __restrict__ counter, unsigned seed) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)G.n * G.n * G.n;
    if (idx >= total) return;
    if (!(G.density[idx] > thres)) return;
    const int have = G.count[idx];
    if (have >= max_ppc) return;
    G.count[idx] = max_ppc;
    const int k = (int)(idx % G.n), j = (int)((idx / G.n) % G.n), i = (int)(idx / ((long)G.n * G.n));
    emit_particles(i, j, k, max_ppc - have, G, out, max_samples, counter, seed);
}

__device__ __forceinline__ void dir_vector(int dir_type, int d[3]) {
    d[0] = d[1] = d[2] = 0;
    if (dir_type >= 0 && dir_type <= 5) d[dir_type >> 1] = (dir_type & 1) ? -1 : 1;
}
// collision_search, filling.py:124-149
__device__ __forceinline__ bool collision_search(const FillGrid& G, int i, int j, int k, int dir_type, float threshold) {
    int d[3];
    dir_vector(dir_type, d);
    i += d[0]; j += d[1]; k += d[2];
    while (max(max(i, j), k) < G.n && min(min(i, j), k) >= 0) {
        if (G.density[((size_t)i * G.n + j) * G.n + k] > threshold) return true;
        i += d[0]; j += d[1]; k += d[2];
    }
    return false;
}
// collision_times, filling.py:152-183
__device__ __forceinline__ int collision_times(const FillGrid& G, int i, int j, int k, int dir_type, float threshold) {
    if (dir_type > 5 || dir_type < 0) return 1;
    int d[3];
    dir_vector(dir_type, d);
    int times = 0;
    bool state = G.count[((size_t)i * G.n + j) * G.n + k] > 0;
    i += d[0]; j += d[1]; k += d[2];
    while (max(max(i, j), k) < G.n && min(min(i, j), k) >= 0) {
        const bool new_state = G.density[((size_t)i * G.n + j) * G.n + k] > threshold;
        if (new_state != state && !state) ++times;
        state = new_state;
        i += d[0]; j += d[1]; k += d[2];
    }
    return times;
}
// internal_filling, filling.py:186-244.  `empty` is a snapshot of (count == 0) taken before the launch: in the reference the
// loop reads grid[i,j,k] == 0 for its own cell only, and a cell filled by another iteration is never revisited.
__global__ __launch_bounds__(256) void internal_fill_kernel(FillGrid G, int max_ppc, int exclude_dir, int ray_cast_dir, float threshold,
                                                            float* __restrict__ out, long long max_samples, unsigned long long* __restrict__ counter,
                                                            unsigned seed) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)G.n * G.n * G.n;
    if (idx >= total) return;
    if (G.count[idx] != 0) return;
    const int k = (int)(idx % G.n), j = (int)((idx / G.n) % G.n), i = (int)(idx / ((long)G.n * G.n));
    bool hit = true;
    for (int dir = 0; dir < 6 && hit; ++dir)
        if (dir != exclude_dir) hit = hit && collision_search(G, i, j, k, dir, threshold);
    if (!hit) return;
    if ((collision_times(G, i, j, k, ray_cast_dir, threshold) & 1) != 1) return;
    G.count[idx] = max_ppc;
    emit_particles(i, j, k, max_ppc, G, out, max_samples, counter, seed ^ 0x5bd1e995u);
}

// assign_particle_to_grid + compute_particle_volume, filling.py:247-288
__global__ __launch_bounds__(256) void volume_count_kernel(const float* __restrict__ pos, int n, FillGrid G) {
    const int pi = blockIdx.x * 256 + threadIdx.x;
    if (pi >= n) return;
    const int i = (int)floorf(pos[3 * pi] / G.dx), j = (int)floorf(pos[3 * pi + 1] / G.dx), k = (int)floorf(pos[3 * pi + 2] / G.dx);
    if ((unsigned)i < (unsigned)G.n && (unsigned)j < (unsigned)G.n && (unsigned)k < (unsigned)G.n)
        atomicAdd(&G.count[((size_t)i * G.n + j) * G.n + k], 1);
}
__global__ __launch_bounds__(256) void volume_kernel(const float* __restrict__ pos, int n, FillGrid G, float* __restrict__ vol) {
    const int pi = blockIdx.x * 256 + threadIdx.x;
    if (pi >= n) return;
    const int i = (int)floorf(pos[3 * pi] / G.dx), j = (int)floorf(pos[3 * pi + 1] / G.dx), k = (int)floorf(pos[3 * pi + 2] / G.dx);
    int c = 1;
    if ((unsigned)i < (unsigned)G.n && (unsigned)j < (unsigned)G.n && (unsigned)k < (unsigned)G.n) c = G.count[((size_t)i * G.n + j) * G.n + k];
    vol[pi] = (G.dx * G.dx * G.dx) / (float)c;
}

// get_attr_from_closest, filling.py:383-403: index of the nearest original particle (first minimum in index order).  The
// original positions pass through LDS in tiles of 1024; one thread per new particle.
__global__ __launch_bounds__(256) void nearest_kernel(const float* __restrict__ pos, int n, const float* __restrict__ new_pos, int n_new,
                                                      int* __restrict__ nearest) {
    __shared__ float tile[1024 * 3];
    const int pi = blockIdx.x * 256 + threadIdx.x;
    float p[3] = {0.f, 0.f, 0.f};
    if (pi < n_new) { p[0] = new_pos[3 * pi]; p[1] = new_pos[3 * pi + 1]; p[2] = new_pos[3 * pi + 2]; }
    float best = 1e10f;
    int best_idx = -1;
    for (int base = 0; base < n; base += 1024) {
        const int m = min(1024, n - base);
        __syncthreads();
        for (int t = threadIdx.x; t < 3 * m; t += 256) tile[t] = pos[3 * (size_t)base + t];
        __syncthreads();
        for (int q = 0; q < m; ++q) {
            const float dx = p[0] - tile[3 * q], dy = p[1] - tile[3 * q + 1], dz = p[2] - tile[3 * q + 2];
            const float dist = sqrtf(dx * dx + dy * dy + dz * dz);   // (p - q).norm(), compared with a strict <
            if (dist < best) { best = dist; best_idx = base + q; }
        }
    }
    if (pi < n_new) nearest[pi] = best_idx;
}

}  // namespace pixie

using namespace pixie;

extern "C" int pixie_fill_densify(const float* d_pos, const float* d_opacity, const float* d_cov6, int n, int grid_n, double grid_dx,
                                  int32_t* d_grid_count, float* d_grid_density, void* stream) {
    PX_REQUIRE(n >= 0 && grid_n > 0 && grid_dx > 0, "pixie_fill_densify: bad arguments");
    if (n == 0) return 0;      // (an empty device array has no address: checked before the pointers)
    PX_REQUIRE(d_pos && d_opacity && d_cov6 && d_grid_count && d_grid_density, "pixie_fill_densify: null argument");
    FillGrid G{grid_n, (float)grid_dx, d_grid_count, d_grid_density};
    hipLaunchKernelGGL(densify_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, as_stream(stream), d_pos, d_opacity, d_cov6, n, G);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int pixie_fill_dense_cells(int32_t* d_grid_count, const float* d_grid_density, int grid_n, double grid_dx, double density_thres,
                                      int max_particles_per_cell, float* d_new_particles, int64_t max_samples, uint64_t* d_counter, uint32_t seed,
                                      void* stream) {
    PX_REQUIRE(d_grid_count && d_grid_density && d_new_particles && d_counter && grid_n > 0 && max_particles_per_cell > 0, "pixie_fill_dense_cells: bad arguments");
    FillGrid G{grid_n, (float)grid_dx, d_grid_count, const_cast<float*>(d_grid_density)};
    const long total = (long)grid_n * grid_n * grid_n;
    hipLaunchKernelGGL(fill_dense_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), G, (float)density_thres,
                       max_particles_per_cell, d_new_particles, (long long)max_samples, reinterpret_cast<unsigned long long*>(d_counter), seed);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int pixie_fill_internal_cells(int32_t* d_grid_count, const float* d_grid_density, int grid_n, double grid_dx, int max_particles_per_cell,
                                         int exclude_dir, int ray_cast_dir, double threshold, float* d_new_particles, int64_t max_samples,
                                         uint64_t* d_counter, uint32_t seed, void* stream) {
    PX_REQUIRE(d_grid_count && d_grid_density && d_new_particles && d_counter && grid_n > 0 && max_particles_per_cell > 0, "pixie_fill_internal_cells: bad arguments");
    FillGrid G{grid_n, (float)grid_dx, d_grid_count, const_cast<float*>(d_grid_density)};
    const long total = (long)grid_n * grid_n * grid_n;
    hipLaunchKernelGGL(internal_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), G, max_particles_per_cell,
                       exclude_dir, ray_cast_dir, (float)threshold, d_new_particles, (long long)max_samples,
                       reinterpret_cast<unsigned long long*>(d_counter), seed);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int pixie_particle_volume(const float* d_pos, int n, int grid_n, double grid_dx, int32_t* d_grid_count_scratch, float* d_vol, void* stream) {
    PX_REQUIRE(d_pos && d_grid_count_scratch && d_vol && n > 0 && grid_n > 0 && grid_dx > 0, "pixie_particle_volume: bad arguments");
    hipStream_t st = as_stream(stream);
    FillGrid G{grid_n, (float)grid_dx, d_grid_count_scratch, nullptr};
    PX_CHECK_HIP(hipMemsetAsync(d_grid_count_scratch, 0, (size_t)grid_n * grid_n * grid_n * sizeof(int32_t), st));
    hipLaunchKernelGGL(volume_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_pos, n, G);
    hipLaunchKernelGGL(volume_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_pos, n, G, d_vol);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int pixie_nearest_particle(const float* d_pos, int n, const float* d_new_pos, int n_new, int32_t* d_nearest, void* stream) {
    PX_REQUIRE(n > 0 && n_new >= 0, "pixie_nearest_particle: bad arguments");
    if (n_new == 0) return 0;
    PX_REQUIRE(d_pos && d_new_pos && d_nearest, "pixie_nearest_particle: null argument");
    hipLaunchKernelGGL(nearest_kernel, dim3((unsigned)((n_new + 255) / 256)), dim3(256), 0, as_stream(stream), d_pos, n, d_new_pos, n_new, d_nearest);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}
