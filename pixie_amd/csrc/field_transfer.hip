// pixie_amd/csrc/field_transfer.hip -- predicted material field -> per-particle material properties on MI355X.
//
// Replaces, between the two halves of the hot path, what the reference does with a PLY file,
// sklearn and a Python loop over every particle:
//   unscale_prediction            pixie/voxel/map_pred_to_coords.py:41-75
//   map_pred_to_ply point list    pixie/voxel/map_pred_to_coords.py:192-252 (masked voxels at np.linspace coordinates,
//                                 material id = argmax of the class channels, conf = their max)
//   perform_knn_smoothing         third_party/PhysGaussian/material_field.py:228-300 (K nearest material points per particle;
//                                 MaterialProperties.assign_from_neighbors :52-78 -- mean / mode, or inverse-distance
//                                 weighted -- and .get_defaults :38-50 for particles whose NEAREST point is too far)
// The material points sit on the voxel lattice, so no spatial index is built: each particle walks the lattice outwards
// in cubic shells from its own voxel, keeping the K best (distance, voxel) pairs, and stops when the next shell cannot
// beat the K-th.  One thread per particle; the un-scaling is applied to the K winners only.  All distances in fp64
// (sklearn computes them in fp64 from the fp32 coordinates).  Nothing touches the host.
#include <hip/hip_runtime.h>

#include "../../include/pixie_hip.h"
#include "common.h"

namespace pixie {

constexpr int kMaxK = 16;

struct FieldArgs {
    const float* pred;          // [3 + ncls][D][H][W]
    const unsigned char* mask;  // [D][H][W], > 0 = occupied
    const float* ax; const float* ay; const float* az;   // voxel coordinates per axis (float32(np.linspace))
    int ncls, D, H, W;
    double hmin;                // smallest lattice spacing
    float dmin, dmax, emin, emax, numin, numax;  // log10 density / log10 E / nu ranges
    const float* pos; int n;    // particles [n][3], in the field's frame
    int k; double thr; int weighted;
    int def_material, def_label;
    const double* sums;         // [4] sums of density, E, nu, conf over the occupied voxels, [4] = their count
    float* density; float* E; float* nu; float* conf; int* material; int* label; float* nearest;
    unsigned long long* n_far;
};

__device__ __forceinline__ float unscale_log(float c, float lo, float hi) {
    c = fminf(fmaxf(c, -1.0f), 1.0f);
    const float l = (c + 1.0f) * (hi - lo) / 2.0f + lo;   // float32 arithmetic, as numpy does on the float32 array
    return powf(10.0f, l);
}
__device__ __forceinline__ float unscale_lin(float c, float lo, float hi) {
    c = fminf(fmaxf(c, -1.0f), 1.0f);
    return (c + 1.0f) * (hi - lo) / 2.0f + lo;
}
__device__ __forceinline__ void voxel_props(const FieldArgs& A, long v, float& dens, float& e, float& nu, int& mid, float& conf) {
    const long S = (long)A.D * A.H * A.W;
    dens = unscale_log(A.pred[v], A.dmin, A.dmax);
    e = unscale_log(A.pred[S + v], A.emin, A.emax);
    nu = unscale_lin(A.pred[2 * S + v], A.numin, A.numax);
    mid = 0;
    conf = A.pred[3 * S + v];
    for (int c = 1; c < A.ncls; ++c) {   // np.argmax: first maximum
        const float p = A.pred[(3 + c) * S + v];
        if (p > conf) { conf = p; mid = c; }
    }
    if (A.ncls <= 1) { mid = (int)A.pred[3 * S + v]; conf = 1.0f; }   // a single class channel IS the class index (get_mat_id, map_pred_to_coords.py:122-126)
}

// numpy's float32 pairwise sum for n <= 128 (what np.mean does on a float32 array): 8 running sums, then the tail
__device__ __forceinline__ float np_sum_f32(const float* a, int n) {
    if (n < 8) {
        float s = 0.0f;
        for (int i = 0; i < n; ++i) s += a[i];
        return s;
    }
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

// per-channel sums over the occupied voxels (for MaterialProperties.get_defaults' np.mean)
__global__ __launch_bounds__(256) void field_defaults_kernel(FieldArgs A, double* __restrict__ sums) {
    const long S = (long)A.D * A.H * A.W;
    double s[5] = {0, 0, 0, 0, 0};
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < S; v += (long)gridDim.x * 256) {
        if (A.mask[v] == 0) continue;
        float d, e, nu, cf; int mid;
        voxel_props(A, v, d, e, nu, mid, cf);
        s[0] += d; s[1] += e; s[2] += nu; s[3] += cf; s[4] += 1.0;
    }
    for (int q = 0; q < 5; ++q) {
        for (int off = 32; off > 0; off >>= 1) s[q] += __shfl_down(s[q], off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&sums[q], s[q]);
    }
}

__global__ __launch_bounds__(128) void field_knn_kernel(FieldArgs A) {
    const int p = blockIdx.x * 128 + threadIdx.x;
    if (p >= A.n) return;
    const double px = A.pos[3 * p], py = A.pos[3 * p + 1], pz = A.pos[3 * p + 2];
    // nearest lattice index per axis (axes are monotone np.linspace arrays)
    auto nearest_idx = [](const float* ax, int n, double x) {
        if (n == 1) return 0;
        const double a0 = ax[0], h = ((double)ax[n - 1] - a0) / (n - 1);
        int i = (int)floor((x - a0) / h + 0.5);
        return i < 0 ? 0 : (i >= n ? n - 1 : i);
    };
    const int cx = nearest_idx(A.ax, A.D, px), cy = nearest_idx(A.ay, A.H, py), cz = nearest_idx(A.az, A.W, pz);
    // how far (in lattice steps) the particle is from that voxel: 0.5 inside the lattice, more outside it
    const double off = fmax(fmax(fabs(px - A.ax[cx]), fabs(py - A.ay[cy])), fabs(pz - A.az[cz]));

    double bd[kMaxK];
    long bi[kMaxK];
    int cnt = 0;
    const int K = A.k;
    // Fewer occupied voxels than K in the whole field (an empty or nearly empty mask): no particle can collect K neighbours, and
    // looking for them would walk the entire lattice once per particle (2 M voxels x 100 k particles at 128^3).  sums[4] = the number
    // of occupied voxels (field_defaults_kernel, the launch before this one): everyone takes the defaults below at once.
    // (sklearn raises there -- "Expected n_neighbors <= n_samples_fit" -- and so does pixie_amd.material_field's host-level entry.)
    const int rmax = (A.sums[4] >= (double)K) ? max(max(A.D, A.H), A.W) : -1;
    for (int r = 0; r <= rmax; ++r) {
        for (int ix = max(cx - r, 0); ix <= min(cx + r, A.D - 1); ++ix) {
            const int adx = abs(ix - cx);
            const double dx = (double)A.ax[ix] - px;
            for (int iy = max(cy - r, 0); iy <= min(cy + r, A.H - 1); ++iy) {
                const int ady = abs(iy - cy);
                const double dy = (double)A.ay[iy] - py;
                const bool face = (adx == r) || (ady == r);
                // on a face of the shell in x or y: the whole z run; otherwise only the two z end points
                const int z0 = max(cz - r, 0), z1 = min(cz + r, A.W - 1);
                const int step = face ? 1 : max(2 * r, 1);
                for (int iz = cz - r; iz <= cz + r; iz += step) {
                    if (iz < z0 || iz > z1) continue;
                    const long v = ((long)ix * A.H + iy) * A.W + iz;
                    if (A.mask[v] == 0) continue;
                    const double dz = (double)A.az[iz] - pz;
                    const double d2 = dx * dx + dy * dy + dz * dz;
                    if (cnt == K && !(d2 < bd[K - 1])) continue;
                    int j = (cnt < K) ? cnt : K - 1;      // insertion, stable for equal distances
                    while (j > 0 && d2 < bd[j - 1]) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; --j; }
                    bd[j] = d2; bi[j] = v;
                    if (cnt < K) ++cnt;
                }
            }
        }
        // every voxel of shell r+1 is at least (r + 1) * hmin - off away
        const double reach = (double)(r + 1) * A.hmin - off;
        if (cnt == K && reach > 0.0 && reach * reach >= bd[K - 1]) break;
    }

    const double nearest = cnt > 0 ? sqrt(bd[0]) : 1e300;
    A.nearest[p] = (float)nearest;
    if (cnt < K || nearest > A.thr) {   // MaterialProperties.get_defaults (:38-50)
        atomicAdd(A.n_far, 1ull);
        const double c = A.sums[4] > 0.0 ? A.sums[4] : 1.0;
        A.density[p] = (float)(A.sums[0] / c); A.E[p] = (float)(A.sums[1] / c); A.nu[p] = (float)(A.sums[2] / c);
        A.conf[p] = (float)(A.sums[3] / c);
        A.material[p] = A.def_material; A.label[p] = A.def_label;
        return;
    }
    float dens[kMaxK], ee[kMaxK], nn[kMaxK], cf[kMaxK];
    int mid[kMaxK];
    for (int j = 0; j < K; ++j) voxel_props(A, bi[j], dens[j], ee[j], nn[j], mid[j], cf[j]);
    double w[kMaxK];
    if (A.weighted) {
        double ws = 0.0;
        for (int j = 0; j < K; ++j) { w[j] = 1.0 / (sqrt(bd[j]) + 1e-8); ws += w[j]; }
        for (int j = 0; j < K; ++j) w[j] /= ws;
        double a = 0, b = 0, c = 0, d = 0;
        for (int j = 0; j < K; ++j) { a += w[j] * dens[j]; b += w[j] * ee[j]; c += w[j] * nn[j]; d += w[j] * cf[j]; }
        A.density[p] = (float)a; A.E[p] = (float)b; A.nu[p] = (float)c; A.conf[p] = (float)d;
    } else {
        const float fk = (float)K;
        A.density[p] = np_sum_f32(dens, K) / fk; A.E[p] = np_sum_f32(ee, K) / fk;
        A.nu[p] = np_sum_f32(nn, K) / fk; A.conf[p] = np_sum_f32(cf, K) / fk;
    }
    // mode: Counter.most_common(1) -> among the most frequent ids the one met first in distance order;
    // weighted: np.unique + bincount(weights) + argmax -> the SMALLEST id among equal vote sums
    int best = mid[0];
    double best_v = -1.0;
    for (int j = 0; j < K; ++j) {
        bool seen = false;
        for (int q = 0; q < j; ++q) seen = seen || (mid[q] == mid[j]);
        if (seen) continue;
        double votes = 0.0;
        for (int q = j; q < K; ++q) if (mid[q] == mid[j]) votes += A.weighted ? w[q] : 1.0;
        const bool better = A.weighted ? (votes > best_v || (votes == best_v && mid[j] < best)) : (votes > best_v);
        if (better) { best_v = votes; best = mid[j]; }
    }
    A.material[p] = best; A.label[p] = best;
}

// unscale_prediction (pixie/voxel/map_pred_to_coords.py:41-75): channels 0..2 clipped to [-1, 1] and mapped back to
// 10^log-range (density, E) / the linear range (nu); the class channels are copied.
__global__ __launch_bounds__(256) void unscale_kernel(const float* __restrict__ pred, float* __restrict__ out, int channels, long S,
                                                      float dmin, float dmax, float emin, float emax, float numin, float numax) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= S) return;
    out[v] = unscale_log(pred[v], dmin, dmax);
    out[S + v] = unscale_log(pred[S + v], emin, emax);
    out[2 * S + v] = unscale_lin(pred[2 * S + v], numin, numax);
    for (int c = 3; c < channels; ++c) out[(long)c * S + v] = pred[(long)c * S + v];
}

// The masked voxel point list of map_pred_to_ply (map_pred_to_coords.py:192-252): one record per voxel with mask > 0, in
// C order of the grid (what numpy's boolean indexing produces).  Stable stream compaction in three launches:
// per-workgroup counts, one scan of those, then each workgroup writes its survivors at base + rank.
constexpr int kPtsWG = 256;
__global__ __launch_bounds__(kPtsWG) void points_count_kernel(const uint8_t* __restrict__ mask, long S, int* __restrict__ counts) {
    const long v = (long)blockIdx.x * kPtsWG + threadIdx.x;
    const int keep = (v < S && mask[v] != 0) ? 1 : 0;
    const int total = __syncthreads_count(keep);
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void points_scan_kernel(int* __restrict__ counts, int nblocks, long long* __restrict__ d_count) {
    __shared__ long long s_sum[1024];
    const int tid = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    const int b0 = min(tid * per, nblocks), b1 = min(b0 + per, nblocks);
    long long sum = 0;
    for (int b = b0; b < b1; ++b) sum += counts[b];
    s_sum[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const long long a = (tid >= off) ? s_sum[tid - off] : 0;
        __syncthreads();
        s_sum[tid] += a;
        __syncthreads();
    }
    long long run = s_sum[tid] - sum;   // exclusive prefix of this thread's chunk
    for (int b = b0; b < b1; ++b) { const int c = counts[b]; counts[b] = (int)run; run += c; }   // (< 2^31 voxels)
    if (tid == 1023) *d_count = s_sum[1023];
}
__global__ __launch_bounds__(kPtsWG) void points_write_kernel(FieldArgs A, const int* __restrict__ offsets, long capacity, float* __restrict__ xyz,
                                                              float* __restrict__ dens, float* __restrict__ E, float* __restrict__ nu,
                                                              int* __restrict__ material, float* __restrict__ conf) {
    __shared__ int s_wave[kPtsWG / 64];
    const long S = (long)A.D * A.H * A.W;
    const long v = (long)blockIdx.x * kPtsWG + threadIdx.x;
    const bool keep = v < S && A.mask[v] != 0;
    const unsigned long long bal = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int base = offsets[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    if (!keep) return;
    const long slot = base + __popcll(bal & ((1ull << lane) - 1ull));
    if (slot >= capacity) return;
    const int iz = (int)(v % A.W), iy = (int)((v / A.W) % A.H), ix = (int)(v / ((long)A.W * A.H));
    xyz[3 * slot] = A.ax[ix]; xyz[3 * slot + 1] = A.ay[iy]; xyz[3 * slot + 2] = A.az[iz];
    float d, e, n, cf; int mid;
    voxel_props(A, v, d, e, n, mid, cf);
    dens[slot] = d; E[slot] = e; nu[slot] = n; material[slot] = mid; conf[slot] = cf;
}

}  // namespace pixie

using namespace pixie;

extern "C" int pixie_unscale_prediction(const float* d_pred, int channels, int64_t spatial, double density_min, double density_max,
                                        double E_min, double E_max, double nu_min, double nu_max, float* d_out, void* stream) {
    PX_REQUIRE(d_pred && d_out && channels >= 3 && spatial > 0, "pixie_unscale_prediction: bad arguments");
    hipLaunchKernelGGL(unscale_kernel, dim3((unsigned)((spatial + 255) / 256)), dim3(256), 0, as_stream(stream), d_pred, d_out, channels,
                       (long)spatial, (float)density_min, (float)density_max, (float)E_min, (float)E_max, (float)nu_min, (float)nu_max);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int64_t pixie_field_points_scratch_bytes(const pixie_field_desc* f) {
    if (!f || f->d <= 0 || f->h <= 0 || f->w <= 0) return 0;
    const long S = (long)f->d * f->h * f->w;
    return (int64_t)(((S + kPtsWG - 1) / kPtsWG) * sizeof(int));
}

extern "C" int pixie_field_points(const pixie_field_desc* f, int64_t capacity, float* d_xyz, float* d_density, float* d_E, float* d_nu,
                                  int32_t* d_material, float* d_conf, int64_t* d_count, void* d_scratch, void* stream) {
    PX_REQUIRE(f && d_count && d_scratch, "pixie_field_points: null argument");
    PX_REQUIRE(f->d_pred && f->d_mask && f->d_axis_x && f->d_axis_y && f->d_axis_z, "pixie_field_points: null field pointer");
    PX_REQUIRE(f->n_classes >= 1 && f->d > 0 && f->h > 0 && f->w > 0, "pixie_field_points: bad field shape");
    PX_REQUIRE(capacity == 0 || (d_xyz && d_density && d_E && d_nu && d_material && d_conf), "pixie_field_points: null output with capacity > 0");
    hipStream_t st = as_stream(stream);
    FieldArgs A{};
    A.pred = f->d_pred; A.mask = f->d_mask; A.ax = f->d_axis_x; A.ay = f->d_axis_y; A.az = f->d_axis_z;
    A.ncls = f->n_classes; A.D = f->d; A.H = f->h; A.W = f->w;
    A.dmin = (float)f->density_min; A.dmax = (float)f->density_max; A.emin = (float)f->E_min; A.emax = (float)f->E_max;
    A.numin = (float)f->nu_min; A.numax = (float)f->nu_max;
    const long S = (long)A.D * A.H * A.W;
    PX_REQUIRE(S < (1l << 31), "pixie_field_points: grid too large");
    const int nblocks = (int)((S + kPtsWG - 1) / kPtsWG);
    int* counts = static_cast<int*>(d_scratch);
    hipLaunchKernelGGL(points_count_kernel, dim3((unsigned)nblocks), dim3(kPtsWG), 0, st, A.mask, S, counts);
    hipLaunchKernelGGL(points_scan_kernel, dim3(1), dim3(1024), 0, st, counts, nblocks, reinterpret_cast<long long*>(d_count));
    if (capacity > 0)
        hipLaunchKernelGGL(points_write_kernel, dim3((unsigned)nblocks), dim3(kPtsWG), 0, st, A, counts, (long)capacity, d_xyz, d_density, d_E,
                           d_nu, d_material, d_conf);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int pixie_field_to_particles(const pixie_field_desc* f, const float* d_pos, int n, int k, double nn_distance_threshold,
                                        int weighted, int default_material, int default_part_label, float* d_density, float* d_E,
                                        float* d_nu, int32_t* d_material, int32_t* d_part_label, float* d_conf, float* d_nearest,
                                        void* d_scratch, void* stream) {
    PX_REQUIRE(f && d_pos && d_density && d_E && d_nu && d_material && d_part_label && d_conf && d_nearest && d_scratch,
               "pixie_field_to_particles: null argument");
    PX_REQUIRE(f->d_pred && f->d_mask && f->d_axis_x && f->d_axis_y && f->d_axis_z, "pixie_field_to_particles: null field pointer");
    PX_REQUIRE(n > 0 && k >= 1 && k <= kMaxK, "pixie_field_to_particles: need n > 0 and 1 <= k <= %d", kMaxK);
    PX_REQUIRE(f->n_classes >= 1 && f->d > 0 && f->h > 0 && f->w > 0, "pixie_field_to_particles: bad field shape");
    hipStream_t st = as_stream(stream);
    FieldArgs A{};
    A.pred = f->d_pred; A.mask = f->d_mask; A.ax = f->d_axis_x; A.ay = f->d_axis_y; A.az = f->d_axis_z;
    A.ncls = f->n_classes; A.D = f->d; A.H = f->h; A.W = f->w;
    A.hmin = f->min_spacing;
    A.dmin = (float)f->density_min; A.dmax = (float)f->density_max; A.emin = (float)f->E_min; A.emax = (float)f->E_max;
    A.numin = (float)f->nu_min; A.numax = (float)f->nu_max;
    A.pos = d_pos; A.n = n; A.k = k; A.thr = nn_distance_threshold; A.weighted = weighted;
    A.def_material = default_material; A.def_label = default_part_label;
    double* sums = static_cast<double*>(d_scratch);               // [5] doubles, then the too-far counter
    A.sums = sums;
    A.n_far = reinterpret_cast<unsigned long long*>(sums + 5);
    A.density = d_density; A.E = d_E; A.nu = d_nu; A.conf = d_conf; A.material = d_material; A.label = d_part_label; A.nearest = d_nearest;
    PX_CHECK_HIP(hipMemsetAsync(d_scratch, 0, 6 * sizeof(double), st));
    const long S = (long)A.D * A.H * A.W;
    hipLaunchKernelGGL(field_defaults_kernel, dim3((unsigned)std::min<long>(1024, (S + 255) / 256)), dim3(256), 0, st, A, sums);
    hipLaunchKernelGGL(field_knn_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, A);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

// ======================================================================================================================
// Density clustering of the "stationary" particles (PhysGaussian/material_field.py:405-406: sklearn.cluster.DBSCAN(eps,
// min_samples).fit_predict) on the device.  The caller bins the points on a lattice of cell size >= eps (a sort by cell
// id -- torch); every point then looks at the 27 cells around its own:
//   core     a point whose closed eps-ball holds >= min_samples points (itself included);
//   clusters connected components of the core points under "within eps" -- lock-free union-find that always links
//            the larger root under the smaller, so a component's root is its LOWEST-INDEX core point, which is also what
//            DBSCAN numbers clusters by;
//   border   a non-core point takes the smallest root among the core points within eps of it; noise gets -1.
// Distances are evaluated in float64 on the float32 coordinates, as scipy / sklearn do.
namespace pixie {

struct DbscanArgs {
    const float* pos;        // [n][3], SORTED by lattice cell
    const int* orig;         // [n]: original index of sorted point s
    const int* cell_start;   // [nx*ny*nz + 1]
    int n, nx, ny, nz;
    float lo[3];
    float inv_cell;
    double eps2;
    int min_samples;
    int* core;               // [n] by sorted index
    int* parent;             // [n] by ORIGINAL index
    int* root;               // [n] by ORIGINAL index: output
};

__device__ __forceinline__ int db_cell(const DbscanArgs& A, float v, int axis, int dim) {
    int c = (int)floorf((v - A.lo[axis]) * A.inv_cell);
    return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

// calls visit(t) for every sorted index t != s whose point lies within eps of point s
template <class F>
__device__ __forceinline__ void db_for_neighbours(const DbscanArgs& A, int s, F visit) {
    const double px = A.pos[3 * s], py = A.pos[3 * s + 1], pz = A.pos[3 * s + 2];
    const int cx = db_cell(A, (float)px, 0, A.nx), cy = db_cell(A, (float)py, 1, A.ny), cz = db_cell(A, (float)pz, 2, A.nz);
    for (int ix = max(cx - 1, 0); ix <= min(cx + 1, A.nx - 1); ++ix)
        for (int iy = max(cy - 1, 0); iy <= min(cy + 1, A.ny - 1); ++iy) {
            // the three z-neighbours are contiguous in the cell order: one range
            const int c0 = (ix * A.ny + iy) * A.nz + max(cz - 1, 0), c1 = (ix * A.ny + iy) * A.nz + min(cz + 1, A.nz - 1);
            for (int t = A.cell_start[c0]; t < A.cell_start[c1 + 1]; ++t) {
                if (t == s) continue;
                const double dx = (double)A.pos[3 * t] - px, dy = (double)A.pos[3 * t + 1] - py, dz = (double)A.pos[3 * t + 2] - pz;
                if (dx * dx + dy * dy + dz * dz <= A.eps2) visit(t);
            }
        }
}

__global__ __launch_bounds__(128) void dbscan_core_kernel(DbscanArgs A) {
    const int s = blockIdx.x * 128 + threadIdx.x;
    if (s >= A.n) return;
    int degree = 1;   // the point itself counts
    db_for_neighbours(A, s, [&](int) { ++degree; });
    A.core[s] = degree >= A.min_samples ? 1 : 0;
    A.parent[A.orig[s]] = A.orig[s];
}

__device__ __forceinline__ int db_find(int* parent, int i) {
    int p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != i) {
        i = p;
        p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return i;
}

__global__ __launch_bounds__(128) void dbscan_union_kernel(DbscanArgs A) {
    const int s = blockIdx.x * 128 + threadIdx.x;
    if (s >= A.n || !A.core[s]) return;
    const int me = A.orig[s];
    db_for_neighbours(A, s, [&](int t) {
        if (!A.core[t]) return;
        int a = me, b = A.orig[t];
        while (true) {   // link the larger root under the smaller one; retry when somebody else moved a root meanwhile
            a = db_find(A.parent, a); b = db_find(A.parent, b);
            if (a == b) break;
            if (a < b) { const int tmp = a; a = b; b = tmp; }
            const int old = atomicMin(&A.parent[a], b);
            if (old == a) break;
            a = old;
        }
    });
}

__global__ __launch_bounds__(128) void dbscan_label_kernel(DbscanArgs A) {
    const int s = blockIdx.x * 128 + threadIdx.x;
    if (s >= A.n) return;
    const int me = A.orig[s];
    if (A.core[s]) { A.root[me] = db_find(A.parent, me); return; }
    int best = 0x7fffffff;
    db_for_neighbours(A, s, [&](int t) { if (A.core[t]) best = min(best, db_find(A.parent, A.orig[t])); });
    A.root[me] = best == 0x7fffffff ? -1 : best;
}

}  // namespace pixie

extern "C" int pixie_dbscan_roots(const float* d_pos_sorted, const int32_t* d_orig_index, const int32_t* d_cell_start, int n, int nx, int ny,
                                  int nz, const double lo[3], double cell_size, double eps, int min_samples, int32_t* d_core_scratch,
                                  int32_t* d_parent_scratch, int32_t* d_root, void* stream) {
    PX_REQUIRE(n >= 0 && nx > 0 && ny > 0 && nz > 0 && eps > 0 && cell_size >= eps && min_samples >= 1 && lo, "pixie_dbscan_roots: bad arguments");
    if (n == 0) return 0;
    PX_REQUIRE(d_pos_sorted && d_orig_index && d_cell_start && d_core_scratch && d_parent_scratch && d_root, "pixie_dbscan_roots: null argument");
    DbscanArgs A{};
    A.pos = d_pos_sorted; A.orig = d_orig_index; A.cell_start = d_cell_start; A.n = n; A.nx = nx; A.ny = ny; A.nz = nz;
    for (int k = 0; k < 3; ++k) A.lo[k] = (float)lo[k];
    A.inv_cell = (float)(1.0 / cell_size);
    A.eps2 = eps * eps;
    A.min_samples = min_samples;
    A.core = d_core_scratch; A.parent = d_parent_scratch; A.root = d_root;
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)((n + 127) / 128)), block(128);
    hipLaunchKernelGGL(dbscan_core_kernel, grid, block, 0, st, A);
    hipLaunchKernelGGL(dbscan_union_kernel, grid, block, 0, st, A);
    hipLaunchKernelGGL(dbscan_label_kernel, grid, block, 0, st, A);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}
