// pixie_amd/csrc/mpm.hip -- MLS-MPM substep for MI355X (gfx950) behind the pixie_mpm_* C ABI.
//
// Replaces the reference's Warp launch sequence MPM_Simulator_WARP.p2g2p
// (third_party/PhysGaussian/mpm_solver_warp/mpm_solver_warp.py:514-637):
//   zero_grid -> pre-P2G modifiers -> compute_stress_from_F_trial -> p2g_apic_with_stress ->
//   grid_normalization_and_gravity -> add_damping_via_grid -> BC collide x k -> g2p
// (6-10 launches + 5 forced device syncs per substep) by TWO launches per substep and no syncs:
//
//   block kernel <G2P,P2G>:  G2P of substep t  ->  modifiers, return map + stress, P2G of
//                            substep t+1.  F_trial, stress, C' and the new v never leave registers
//                            between the gather and the scatter (in the fused form v and C are not
//                            even stored: nothing reads them before the next G2P overwrites them).
//   grid kernel:             normalise + gravity + damping + every BC in one sweep, and
//                            clears (m, m*v) behind itself so no separate zero_grid runs.
//
// Particles are kept binned by 4x4x4-cell grid block (counting sort every `resort_interval`
// substeps, all on the device).  One workgroup serves <= 256 particles of one block: it stages the
// block's 8x8x8 node neighbourhood of grid velocities in LDS (one coalesced pass), every particle
// gathers its 27 nodes from LDS, scatters its 27 x (m*v, m) contributions with LDS float atomics
// (ds_add_f32) into a second LDS tile, and the workgroup flushes the tile's non-zero nodes to HBM with
// one global fp32 atomic per word: ~2k global atomics per workgroup instead of 27.6k.  A particle whose
// stencil has drifted out of its workgroup's tile (stale binning) takes a slow path straight to
// global memory, so correctness never depends on the binning being fresh.
//
// Layout in HBM: particle state is SoA fp32/int32 rows of one [45][n] word array (each row a fully
// coalesced 256 B/wave stream; two copies, ping-ponged by the re-binning permutation); the grid is two
// float4 arrays: gin = (m*v.xyz, m) that P2G accumulates, gout = (v.xyz, 0) that G2P gathers.  `perm`
// maps internal slots to the caller's particle order; every import/export goes through it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/pixie_hip.h"
#include "common.h"
#include "mpm_math.h"

namespace pixie {

constexpr int kBS = 4;                    // cells per block edge (particles are binned by the block of their stencil base)
constexpr int kTS = 8;                    // tile nodes per edge: kBS + 2 (stencil reach) + 2 (one-cell drift margin each side)
constexpr int kTN = kTS * kTS * kTS;      // 512 nodes
static_assert(kTN == 512 && kTS == 8, "gather_node recovers the work item from a tile offset with >> 9 and packs tile coordinates in 3 bits");
constexpr int kWG = 256;                  // particles per work item / threads per workgroup (upper value)

// rows of the particle word array
enum Row {
    R_X = 0, R_V = 3, R_F = 6, R_FT = 15, R_C = 24, R_VOL = 33, R_MASS, R_DENSITY, R_E, R_NU, R_MU, R_LAM, R_BULK, R_YS,
    R_MATERIAL, R_SELECTION, R_PERM, R_XLO, R_XREF = R_XLO + 3, R_COUNT = R_XREF + 3
    // R_XLO: the low-order part of the position (set_scalar "compensated_x": x_true = x + xlo, |xlo| <= ulp(x)/2; zero otherwise);
    // R_XREF: position at the last re-binning (must stay the last rows: bin_permute_kernel fills them from x)
};
static_assert(R_COUNT == 51, "row table");

struct MpmPtrs {
    int n, ng, nbk;   // particles, grid nodes per axis, blocks per axis
    float dx, inv_dx;
    float *x, *v, *F, *Ft, *C;  // SoA: [3][n], [3][n], [9][n], [9][n], [9][n]
    float *vol, *mass, *density, *E, *nu, *mu, *lam, *bulk, *ys;
    int *material, *selection, *perm;
    float* xref;                 // [3][n] positions at the last re-binning (drift measurement)
    float* xlo;                  // [3][n] what the float32 position leaves behind (compensated_x), see particle_phase1
    float4 *gin, *gout;
    const int4* items;           // work list: (block id, first slot, count, 0)
    float4* part;                // [n_items][kTN]: (m*v.xyz, m) of each work item's tile, written by its P2G
    unsigned long long* tile_mask;  // [n_items][8]: bit t of a tile = "node t (tile coordinates, z fastest) is not all zero"
    unsigned zero_off;           // index (float4 units) of an all-zero tile behind the work items' tiles: what a gather reads where there is nothing to read
    int sparse_tiles;            // 1: P2G stores only the non-zero nodes of a tile and the grid kernel reads only those (masks);
                                 // 0: whole tiles both ways (scenes too small to be bandwidth-bound: one dependent load fewer)
    const int2* blk_items;       // per block: (first work item, number of work items)
    int* blk_flags;              // per block: bit 0 = active (particles nearby), bit 1 = slow-path particles wrote into gin here
    const int* active_list;      // the active blocks
    const int2* nbr_table;       // per active block (same order): [0] = (block id, 0), [1..27] = blk_items of its 27 neighbours
    const unsigned* staged_lut;  // [256]: staged_index of tile nodes t (low half) and t + 256 (high half)
    unsigned long long* oob;     // [0] particles skipped because their stencil left the grid, [1] slow-path particles,
                                 // [2] slow-path particles dropped because they had left every active block
};

struct StepParams {
    float dt, time;
    float g[3];
    float damping;
    int do_damping;
    float rpic;
    int xcd_order;  // set_scalar "xcd_order": work items dealt to the XCDs in contiguous runs (see mpm_block_kernel)
    int comp_x;     // set_scalar "compensated_x": carry the rounding error of x += dt v in xlo (see particle_phase1)
    int trace;      // timing studies: bit 0 = workgroups stamp s_memrealtime around their phases into g_mpm_trace;
                    // bits 8.. = ablations for bottleneck hunting (RESULTS ARE WRONG with any of them set):
                    // 0x100 skip the LDS scatter atomics, 0x200 skip the workgroup scale reduction (fixed scale),
                    // 0x400 skip the tile staging loads, 0x800 skip the tile publish stores
    MaterialScalars ms;
};

struct BCDev {
    int type, surface_type, reset, pad;
    float point[3], size[3], velocity[3], normal[3];
    float start, end, friction;
};
constexpr int kMaxBCPerLaunch = 16;
struct BCSet {
    int n;
    BCDev bc[kMaxBCPerLaunch];
};

struct PModDev {
    int type;
    float point[3], force[3], velocity[3], normal[3], h1[3], h2[3];
    float rot_scale, trans_scale, start, end;
    const int* mask;   // [n] in the CALLER's particle order (index through perm)
};
constexpr int kMaxPModFused = 8;
struct PModSet {
    int n;
    PModDev pm[kMaxPModFused];
};

// Position of tile node (lx,ly,lz) inside a STAGED tile (part[item][512]).  A tile starts one node before its block, so
// along each axis its 8 nodes fall into three grid blocks: {0}, {1..4}, {5..7}.  The staged tile is stored sub-box by
// sub-box (27 boxes of 1|4|3 nodes per axis, each contiguous) so that the grid kernel, whose wave owns one 4x4x4 grid
// block, reads ONE contiguous run per covering tile instead of 16 64-byte rows scattered over 128-byte lines
// (measured: 3x over-fetch with the plain row-major tile).
__host__ __device__ __forceinline__ int staged_index(int lx, int ly, int lz) {
    const int sx = (lx == 0) ? 0 : (lx <= 4 ? 1 : 2), sy = (ly == 0) ? 0 : (ly <= 4 ? 1 : 2), sz = (lz == 0) ? 0 : (lz <= 4 ? 1 : 2);
    const int ox = (sx == 0) ? 0 : (sx == 1 ? 1 : 5), oy = (sy == 0) ? 0 : (sy == 1 ? 1 : 5), oz = (sz == 0) ? 0 : (sz == 1 ? 1 : 5);
    const int ny = (sy == 0) ? 1 : (sy == 1 ? 4 : 3), nz = (sz == 0) ? 1 : (sz == 1 ? 4 : 3);
    // nodes in the sub-boxes that precede (sx,sy,sz): full x-slabs, then full y-rows of this slab, then z-boxes of this row
    const int nx = (sx == 0) ? 1 : (sx == 1 ? 4 : 3);
    const int before = ox * 64 + nx * (oy * 8 + ny * oz);
    return before + ((lx - ox) * ny + (ly - oy)) * nz + (lz - oz);
}

// ------------------------------------------------------------------ particle modifiers
// apply_force (mpm_solver_warp.py:1015-1027), modify_particle_v_before_p2g (:1061-1073, :1137-1179)
__device__ __forceinline__ void apply_pmod(const PModDev& m, int caller_idx, float time, float dt, float mass,
                                           const float x[3], float v[3]) {
    if (!(time >= m.start && time < m.end)) return;
    if (m.mask[caller_idx] != 1) return;
    if (m.type == PIXIE_PM_IMPULSE) {
        for (int d = 0; d < 3; ++d) v[d] = v[d] + (m.force[d] / mass) * dt;
    } else if (m.type == PIXIE_PM_TRANSLATION) {
        for (int d = 0; d < 3; ++d) v[d] = m.velocity[d];
    } else {
        const float o[3] = {x[0] - m.point[0], x[1] - m.point[1], x[2] - m.point[2]};
        const float dn = o[0] * m.normal[0] + o[1] * m.normal[1] + o[2] * m.normal[2];
        const float hx = o[0] - dn * m.normal[0], hy = o[1] - dn * m.normal[1], hz = o[2] - dn * m.normal[2];
        const float hd = sqrtf(hx * hx + hy * hy + hz * hz);
        const float cosine = (o[0] * m.h1[0] + o[1] * m.h1[1] + o[2] * m.h1[2]) / hd;
        float theta = acosf(cosine);
        if (!((o[0] * m.h2[0] + o[1] * m.h2[1] + o[2] * m.h2[2]) > 0.0f)) theta = -theta;
        const float a1 = -hd * sinf(theta) * m.rot_scale;
        const float a2 = hd * cosf(theta) * m.rot_scale;
        for (int d = 0; d < 3; ++d) v[d] = a1 * m.h1[d] + a2 * m.h2[d] + m.trans_scale * m.normal[d];
    }
}

__device__ __forceinline__ bool stencil_inside(const Stencil& st, int ng) {
    bool ok = true;
    for (int d = 0; d < 3; ++d) ok = ok && (st.base[d] >= 0) && (st.base[d] + 2 < ng);
    return ok;
}

// ------------------------------------------------------------------ separable transfer cores
// The 27-node sums of g2p (mpm_utils.py:436-455) and p2g_apic_with_stress (:360-393) are tensor products of
// 1-D quadratic B-spline weights, so they are evaluated axis by axis (z innermost): 9 FMAs per node
// instead of ~28.  This only re-associates the reference's fp32 sums.
struct Weights1D {
    float w[3], dw[3], wd[3];  // weight, derivative (cell units), weight * (offset - fx)
};
__device__ __forceinline__ void weights_1d(const Stencil& st, int d, Weights1D& o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o.w[i] = st.w[d][i];
        o.dw[i] = st.dw[d][i];
        o.wd[i] = st.w[d][i] * ((float)i - st.fx[d]);
    }
}

// v = sum w g;  B_ab = sum w g_a dpos_b (cell units);  G_ab = sum g_a dweight_b (cell units)
// Two contractions of every node value g_a(i,j,k) carry all 21 sums (6 FMAs per node instead of 9; 288 VALU instructions per
// particle instead of 441 -- round 4, the kernel is VALU-issue bound):
//   s_a(i,j) = sum_k wz_k g_a       -> the x- and y-derivative columns:  ss_a(i) = sum_j wy_j s_a,  t_a(j) = sum_i wx_i s_a
//   h_a(k)   = sum_ij wx_i wy_j g_a -> v and the z-derivative column
//   v_a = sum_k wz_k h_a(k);   G_a0 = sum_i dwx_i ss_a(i),  G_a1 = sum_j dwy_j t_a(j),  G_a2 = sum_k dwz_k h_a(k);   B likewise with wd.
// SCHED: keep the loads of one x-slab from being hoisted over the previous slab's arithmetic (the 5-waves-per-SIMD register
// budget needs it; the wide variant for small scenes, F_WIDE, lets the compiler overlap everything).
template <bool SCHED, class Fetch>
__device__ __forceinline__ void g2p_gather(const Stencil& st, Fetch fetch, float nv[3], Mat3& B, Mat3& G) {
    Weights1D wx, wy, wz;
    weights_1d(st, 0, wx); weights_1d(st, 1, wy); weights_1d(st, 2, wz);
    float h[3][3], t[3][3], g0[3], b0[3];   // h[k][a], t[j][a]
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        g0[a] = 0.0f; b0[a] = 0.0f;
#pragma unroll
        for (int o = 0; o < 3; ++o) { h[o][a] = 0.0f; t[o][a] = 0.0f; }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float ss[3] = {0, 0, 0};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float wij = wx.w[i] * wy.w[j];
            float s[3] = {0, 0, 0};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float g[3];
                fetch(i, j, k, g);
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    s[a] = fmaf(wz.w[k], g[a], s[a]);
                    h[k][a] = fmaf(wij, g[a], h[k][a]);
                }
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                ss[a] = fmaf(wy.w[j], s[a], ss[a]);
                t[j][a] = fmaf(wx.w[i], s[a], t[j][a]);
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            g0[a] = fmaf(wx.dw[i], ss[a], g0[a]);
            b0[a] = fmaf(wx.wd[i], ss[a], b0[a]);
        }
        if (SCHED) __builtin_amdgcn_sched_barrier(0);  // keep the 27 loads of the next x-slab from being hoisted over this one
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        nv[a] = fmaf(wz.w[2], h[2][a], fmaf(wz.w[1], h[1][a], wz.w[0] * h[0][a]));
        G.m[3 * a + 0] = g0[a];
        B.m[3 * a + 0] = b0[a];
        G.m[3 * a + 1] = fmaf(wy.dw[2], t[2][a], fmaf(wy.dw[1], t[1][a], wy.dw[0] * t[0][a]));
        B.m[3 * a + 1] = fmaf(wy.wd[2], t[2][a], fmaf(wy.wd[1], t[1][a], wy.wd[0] * t[0][a]));
        G.m[3 * a + 2] = fmaf(wz.dw[2], h[2][a], fmaf(wz.dw[1], h[1][a], wz.dw[0] * h[0][a]));
        B.m[3 * a + 2] = fmaf(wz.wd[2], h[2][a], fmaf(wz.wd[1], h[1][a], wz.wd[0] * h[0][a]));
    }
}

// momentum_a(i,j,k) = w (mv_a + A_a . dpos) + T_a . gradw,  mass(i,j,k) = w m   with A = m C' dx (dpos in cell
// units) and T = -dt vol inv_dx tau (gradw in cell units).  Grouped by axis (wd = w * offset, dw = dw/dx in cell units):
//   momentum_a = wz_k [ wy_j E_a(i) + wx_i U_a(j) ] + wx_i wy_j S_a(k)
//   E_a(i) = wx_i mv_a + A_a0 wdx_i + T_a0 dwx_i,   U_a(j) = A_a1 wdy_j + T_a1 dwy_j,   S_a(k) = A_a2 wdz_k + T_a2 dwz_k
// U and S do not depend on the other axes and are formed once per particle: 7 VALU instructions per node, 8 per (i, j)
// pair, 9 per x-slab, 36 once -- 324 per particle (round 3: 486; the block kernel is VALU-issue bound).
template <bool SCHED, class Emit>
__device__ __forceinline__ void p2g_scatter(const Stencil& st, const float mv[3], const Mat3& A, const Mat3& T, float mass, Emit emit) {
    Weights1D wx, wy, wz;
    weights_1d(st, 0, wx); weights_1d(st, 1, wy); weights_1d(st, 2, wz);
    float U[3][3], S[3][3];   // [component][offset]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            U[a][o] = fmaf(A.m[3 * a + 1], wy.wd[o], T.m[3 * a + 1] * wy.dw[o]);
            S[a][o] = fmaf(A.m[3 * a + 2], wz.wd[o], T.m[3 * a + 2] * wz.dw[o]);
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float E[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) E[a] = fmaf(A.m[3 * a + 0], wx.wd[i], fmaf(T.m[3 * a + 0], wx.dw[i], wx.w[i] * mv[a]));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float wij = wx.w[i] * wy.w[j];
            float P[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) P[a] = fmaf(E[a], wy.w[j], wx.w[i] * U[a][j]);
            const float M = wij * mass;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float mom[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) mom[a] = fmaf(wz.w[k], P[a], wij * S[a][k]);
                emit(i, j, k, mom, wz.w[k] * M);
            }
        }
        if (SCHED) __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- slow path: a particle whose stencil lies outside its workgroup's tile talks to HBM directly.  Rare
// (stale binning only), so it is kept out of line and rolled to stay out of the fast path's register budget.
__device__ __noinline__ void g2p_gather_global(const float4* __restrict__ gout, int ng, Stencil st, float* nv9 /* nv[3], B[9], G[9] */) {
    float nv[3] = {0, 0, 0}, B[9], G[9];
    for (int q = 0; q < 9; ++q) { B[q] = 0.0f; G[q] = 0.0f; }
    const size_t r0 = ((size_t)st.base[0] * ng + st.base[1]) * ng + st.base[2];
#pragma unroll 1
    for (int t = 0; t < 27; ++t) {
        const int i = t / 9, j = (t / 3) % 3, k = t % 3;
        const float4 q = gout[r0 + ((size_t)i * ng + j) * ng + k];
        const float g[3] = {q.x, q.y, q.z};
        const float w = st.w[0][i] * st.w[1][j] * st.w[2][k];
        const float gw[3] = {st.dw[0][i] * st.w[1][j] * st.w[2][k], st.w[0][i] * st.dw[1][j] * st.w[2][k],
                             st.w[0][i] * st.w[1][j] * st.dw[2][k]};
        const float dp[3] = {(float)i - st.fx[0], (float)j - st.fx[1], (float)k - st.fx[2]};
        for (int a = 0; a < 3; ++a) {
            nv[a] += w * g[a];
            for (int b = 0; b < 3; ++b) {
                B[3 * a + b] += w * g[a] * dp[b];
                G[3 * a + b] += g[a] * gw[b];
            }
        }
    }
    for (int a = 0; a < 3; ++a) nv9[a] = nv[a];
    for (int q = 0; q < 9; ++q) { nv9[3 + q] = B[q]; nv9[12 + q] = G[q]; }
}

__device__ __noinline__ bool p2g_scatter_global(float4* gin, int* blk_flags, int nbk, int ng, Stencil st,
                                                const float* mvAT /* mv[3], A[9], T[9] */, float mass) {
    const size_t r0 = ((size_t)st.base[0] * ng + st.base[1]) * ng + st.base[2];
    // the 3x3x3 stencil touches at most 2 blocks per axis: tell the grid kernel they hold gin contributions.  The grid
    // kernel only visits ACTIVE blocks (particles within one block at the last re-binning); a particle that has left
    // all of them drifted >= 3 cells since then although the re-binning cadence is set to keep the drift below half a
    // cell (rebin()) -- it is dropped and counted, like a particle that leaves the grid.
    bool reachable = true;
    for (int c = 0; c < 8; ++c) {
        const int bx = (st.base[0] + 2 * (c >> 2)) / kBS, by = (st.base[1] + 2 * ((c >> 1) & 1)) / kBS, bz = (st.base[2] + 2 * (c & 1)) / kBS;
        reachable = reachable && (atomicOr(&blk_flags[(bx * nbk + by) * nbk + bz], 2) & 1);
    }
    if (!reachable) return false;
#pragma unroll 1
    for (int t = 0; t < 27; ++t) {
        const int i = t / 9, j = (t / 3) % 3, k = t % 3;
        const float w = st.w[0][i] * st.w[1][j] * st.w[2][k];
        const float gw[3] = {st.dw[0][i] * st.w[1][j] * st.w[2][k], st.w[0][i] * st.dw[1][j] * st.w[2][k],
                             st.w[0][i] * st.w[1][j] * st.dw[2][k]};
        const float dp[3] = {(float)i - st.fx[0], (float)j - st.fx[1], (float)k - st.fx[2]};
        float* cell = reinterpret_cast<float*>(gin + r0 + ((size_t)i * ng + j) * ng + k);
        for (int a = 0; a < 3; ++a) {
            const float* A = mvAT + 3 + 3 * a;
            const float* T = mvAT + 12 + 3 * a;
            const float mom = w * (mvAT[a] + A[0] * dp[0] + A[1] * dp[1] + A[2] * dp[2]) + T[0] * gw[0] + T[1] * gw[1] + T[2] * gw[2];
            unsafeAtomicAdd(cell + a, mom);
        }
        unsafeAtomicAdd(cell + 3, w * mass);
    }
    return true;
}

// ------------------------------------------------------------------ fused block kernel
// G2P part: g2p (mpm_utils.py:412-463).  P2G part: pre-P2G modifiers, compute_stress_from_F_trial
// (:467-526) and p2g_apic_with_stress (:338-394).  `sp.time` is the time of the substep whose P2G runs.
//
// The scatter accumulates in LDS with 64-bit INTEGER atomics: ds_add_f32 retires ~1 lane per 3 cycles on gfx950
// (193 cycles per wave-instruction, measured: scripts/microbench/lds_atomics.hip), ds_add_u64 is 11x faster.  Each
// workgroup therefore (1) computes every particle's scatter inputs, (2) takes the workgroup maximum of a bound on
// their contributions, (3) scales by the power of two that puts that bound just below 2^42 and adds round-to-nearest
// integers (|sum of 256| < 2^50; see to_fixed).  The LSB is 2^-42 of the largest contribution in the tile, i.e. the tile
// sums are exact to ~2e-13 -- tighter than any fp32 summation order -- and since integer adds commute they are bit-reproducible.
// (A 32-bit variant -- LSB 2^-21 -- was measured first: low-mass free-surface nodes lost up to 10 % of their velocity.)
struct ScatterIn {
    float x[3];     // position after the G2P update (where the P2G stencil is taken)
    float mv[3];    // mass * v
    Mat3 A;         // mass * C' * dx
    Mat3 T;         // -dt * vol * inv_dx * tau
    float mass;
    bool active;    // contributes to P2G
};

// what a particle needs from HBM before it can start: issued BEFORE the tile load + barrier so that the two latencies overlap
struct Preload {
    float x[3];
    Mat3 F;          // fused kernel: F of the previous substep (G2P input); P2G-only kernel: unused
    float mass, vol, mu, lam;
    int material, selection;
};
template <bool DO_G2P, bool DO_P2G>
__device__ __forceinline__ void preload_particle(const MpmPtrs& S, int p, Preload& L) {
    const int n = S.n;
    L.selection = S.selection[p];
#pragma unroll
    for (int d = 0; d < 3; ++d) L.x[d] = S.x[d * n + p];
    if (DO_G2P) {
#pragma unroll
        for (int i = 0; i < 9; ++i) L.F.m[i] = S.F[i * n + p];
    }
    if (DO_P2G) {
        L.mass = S.mass[p]; L.vol = S.vol[p]; L.mu = S.mu[p]; L.lam = S.lam[p]; L.material = S.material[p];
    }
}

// What a caller reads from a particle that has just been frozen outside the grid: zero velocity and APIC matrix, and its last
// deformation gradient as F_trial.  (The fused kernel keeps v, C and F_trial of the particles that take part in registers, and a
// re-binning does not move those rows for them -- bin_permute_kernel -- so the rows of a particle that drops out are written here.)
// (inlined, a rolled loop of plain stores: an out-of-line function taking the kernel's parameter struct by reference copies it to
// scratch -- 404 -> 740 bytes per lane, beyond the cliff of profiles/r6e_scratch_regression.txt)
__device__ __forceinline__ void freeze_particle_state(const MpmPtrs& S, int p, const Mat3& F) {
    const int n = S.n;
#pragma unroll
    for (int d = 0; d < 3; ++d) S.v[d * n + p] = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) { S.C[i * n + p] = 0.0f; S.Ft[i * n + p] = F.m[i]; }
}

template <bool DO_G2P, bool DO_P2G, bool SCHED>
__device__ __forceinline__ void particle_phase1(const MpmPtrs& S, const StepParams& sp, const PModSet& pms, int p, int ox, int oy,
                                                int oz, const float4* tv, const Preload& L, ScatterIn& out) {
    out.active = false;
    if (L.selection != 0) return;
    const int n = S.n;
    float x[3], v[3];
    Mat3 C, Ft;
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = L.x[d];

    if (DO_G2P) {
        const Stencil st = make_stencil(x[0], x[1], x[2], S.inv_dx);
        if (!stencil_inside(st, S.ng)) {
            atomicAdd(S.oob, 1ull);
            S.selection[p] = 2;   // left the grid: frozen from now on and counted once (UB in the reference)
            freeze_particle_state(S, p, L.F);
            return;
        }
        const Mat3& Fold = L.F;
        float nv[3];
        Mat3 B, G;
        const int lx = st.base[0] - ox, ly = st.base[1] - oy, lz = st.base[2] - oz;
        if ((unsigned)lx <= (unsigned)(kTS - 3) && (unsigned)ly <= (unsigned)(kTS - 3) && (unsigned)lz <= (unsigned)(kTS - 3)) {
            const int b0 = (lx * kTS + ly) * kTS + lz;
            g2p_gather<SCHED>(st, [&](int i, int j, int k, float g[3]) {
                // One ds_read_b128 per node.  The .w lane is dead, but a 16-byte LDS read costs 4 LDS cycles per wave against
                // 8 for the 12-byte ds_read_b96 the compiler would narrow it to (MI355X_MICROARCH.md, LDS table), and LDS
                // and VALU time add up in this kernel (80.7 -> 78.4 us per launch at 1 M particles): keep the lane alive.
                const float4 q = tv[b0 + (i * kTS + j) * kTS + k];
                asm volatile("" :: "v"(q.w));
                g[0] = q.x; g[1] = q.y; g[2] = q.z;
            }, nv, B, G);
        } else {
            atomicAdd(S.oob + 1, 1ull);
            float acc[21];
            g2p_gather_global(S.gout, S.ng, st, acc);
#pragma unroll
            for (int a = 0; a < 3; ++a) nv[a] = acc[a];
#pragma unroll
            for (int q = 0; q < 9; ++q) { B.m[q] = acc[3 + q]; G.m[q] = acc[12 + q]; }
        }
        Mat3 Amat;
        const float sdt = sp.dt * S.inv_dx;
#pragma unroll
        for (int i = 0; i < 9; ++i) Amat.m[i] = ((i % 4 == 0) ? 1.0f : 0.0f) + G.m[i] * sdt;
        Ft = mat_mul(Amat, Fold);
        const float sc = 4.0f * S.inv_dx;
#pragma unroll
        for (int i = 0; i < 9; ++i) C.m[i] = B.m[i] * sc;
        // x += dt v in float32 (mpm_utils.py:447) drops whatever of dt v lies below ulp(x)/2 = 6e-8 at x ~ 1: in a quiet scene
        // (dt v ~ 1e-7) most of the motion, systematically -- the reference's own float32 behaviour, and the default here.
        // "compensated_x" keeps what was dropped in xlo and feeds it into the next increment (Kahan): x is then the float32
        // ROUNDING of the accumulated position instead of a sum of rounded increments (3 more words per particle each way).
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            v[d] = nv[d];
            if (sp.comp_x) {
                const float y = sp.dt * nv[d] + S.xlo[d * n + p];
                const float t = x[d] + y;
                S.xlo[d * n + p] = y - (t - x[d]);
                x[d] = t;
            } else {
                x[d] = x[d] + sp.dt * nv[d];
            }
            S.x[d * n + p] = x[d];
        }
        if (!DO_P2G) {  // the state a caller can observe: x, v, C, F_trial
#pragma unroll
            for (int d = 0; d < 3; ++d) S.v[d * n + p] = v[d];
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                S.C[i * n + p] = C.m[i];
                S.Ft[i * n + p] = Ft.m[i];
            }
            return;
        }
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) v[d] = S.v[d * n + p];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            C.m[i] = S.C[i * n + p];
            Ft.m[i] = S.Ft[i * n + p];
        }
    }

    if (DO_P2G) {
        const float mass = L.mass;
        if (pms.n > 0) {
            const float v_before[3] = {v[0], v[1], v[2]};
            const int ci = S.perm[p];
            for (int k = 0; k < pms.n; ++k) apply_pmod(pms.pm[k], ci, sp.time, sp.dt, mass, x, v);
            if (!DO_G2P && (v[0] != v_before[0] || v[1] != v_before[1] || v[2] != v_before[2]))
                for (int d = 0; d < 3; ++d) S.v[d * n + p] = v[d];
        }
        const int material = L.material;
        float mu = L.mu, lam = L.lam;
        float ys = (material == 1 || material == 3 || material == 5) ? S.ys[p] : 0.0f;
        const float bulk = (material == 6) ? S.bulk[p] : 0.0f;
        const float mu0 = mu, lam0 = lam, ys0 = ys;
        Mat3 F, tau;
        return_map_and_stress(material, Ft, mu, lam, bulk, ys, sp.ms, sp.dt, F, tau);
#pragma unroll
        for (int i = 0; i < 9; ++i) S.F[i * n + p] = F.m[i];
        if (ys != ys0) S.ys[p] = ys;
        if (mu != mu0) S.mu[p] = mu;
        if (lam != lam0) S.lam[p] = lam;

        // C' = (1-r) C + r/2 (C - C^T);  r < -0.001 => PIC (mpm_utils.py:372-379)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float c = (1.0f - sp.rpic) * C.m[3 * a + b] + sp.rpic * 0.5f * (C.m[3 * a + b] - C.m[3 * b + a]);
                if (sp.rpic < -0.001f) c = 0.0f;
                out.A.m[3 * a + b] = mass * c * S.dx;  // dpos = (ijk - fx) * dx
            }
        const float ks = -sp.dt * L.vol * S.inv_dx;  // dt * (-vol * tau * dweight), dweight = dw*w*w*inv_dx
#pragma unroll
        for (int i = 0; i < 9; ++i) out.T.m[i] = ks * tau.m[i];
#pragma unroll
        for (int d = 0; d < 3; ++d) { out.mv[d] = mass * v[d]; out.x[d] = x[d]; }
        out.mass = mass;
        out.active = true;
    }
}

// ---- fixed-point accumulation of the scatter ---------------------------------------------------------------------------
// EXACT mode (set_scalar "scatter_bits" 64): 64-bit fixed point through the double-precision adder: the mantissa field of (x + 1.5 * 2^52) is
// 2^51 + round(x) for |x| < 2^51, so ds_add_u64 of the raw bit patterns accumulates sum(round(x_i)) modulo 2^51 in the low
// 51 bits whatever happens above them (the N copies of the exponent and of the 2^51 offset only carry upwards).  With every
// contribution scaled below 2^42 and at most 256 of them per node the sum stays below 2^50 and is recovered by
// sign-extending bit 50: two instructions per contribution (v_cvt_f64_f32, v_add_f64), exact integer accumulation,
// order-independent.
// PACKED mode (F_PACK32, the default; "scatter_bits" 32): two 32-bit two's-complement integers per ds_add_u64 -- (m*v.x, m*v.y)
// and (m, m*v.z) -- so a node costs 2 LDS atomics instead of 4 and 6 plain VALU instructions instead of 8 double-rate ones.
// The contributions are scaled so that the sum of the particles' bounds is below 2^30 (see the scales in the kernel) and
// rounded to nearest by v_cvt_rpi_i32_f32; the
// pair word is low + (high << 32) in 64-bit arithmetic, i.e. the high dword carries `high + (low >> 31)`, and the decode
// undoes exactly that, so the two sums are exact integers and order-independent like the 64-bit ones.  What changes is the
// quantum: 2^-30 of the SUM of the contribution bounds of the tile's particles instead of 2^-42 of their maximum -- the
// order of the fp32 atomics of the reference (2^-24 of each partial sum) for nodes that carry mass, but a node whose whole
// mass is below ~1e-7 of a particle's (stencil corners at a free surface) is quantised visibly: see HISTORY.md 3.5 and
// tests/test_mpm_hip.py::test_packed_scatter_parity.
constexpr double kMagicD = 6755399441055744.0;            // 1.5 * 2^52

// power of two s with bound * s in [2^top, 2^(top+1))  (1 when the bound is zero / not finite)
__device__ __forceinline__ float scale_for(float bound, int top) {
    const unsigned bits = __float_as_uint(bound);
    const int eb = (int)((bits >> 23) & 0xffu) - 127;
    if (!(bound > 0.0f) || eb > 120) return 1.0f;
    int e = top - eb;
    e = e > 120 ? 120 : (e < -80 ? -80 : e);
    return __uint_as_float((unsigned)(e + 127) << 23);
}
// 1 / s for s = 2^e, exactly, without a division (|e| <= 120: scale_for)
__device__ __forceinline__ float pow2_reciprocal(float s) {
    return __uint_as_float(0x7f000000u - __float_as_uint(s));
}
__device__ __forceinline__ unsigned long long to_fixed(float scaled) {
    return (unsigned long long)__double_as_longlong((double)scaled + kMagicD);
}
// low 51 bits, sign-extended, as a float: flipping bit 50 turns the field into (sum + 2^50) >= 0, which is dropped into the
// mantissa of 2^52 and the two offsets subtracted again -- exact (|sum| < 2^50), and no 64-bit integer conversion
__device__ __forceinline__ float from_fixed(unsigned long long v, float inv_scale) {
    const unsigned hi = (((unsigned)(v >> 32) & 0x7ffffu) ^ 0x40000u) | 0x43300000u;
    const double d = __hiloint2double((int)hi, (int)(unsigned)v) - 5629499534213120.0;   // 2^52 + 2^50
    return (float)(d * (double)inv_scale);
}
__device__ __forceinline__ int round_to_int(float x) {   // floor(x + 0.5): one instruction (CDNA keeps GCN's v_cvt_rpi)
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ int round_to_int_neg(float x) {   // floor(-x + 0.5): the negation rides in the VOP3 source modifier
    int r;
    asm("v_cvt_rpi_i32_f32_e64 %0, -%1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ unsigned long long pack_pair(int low, int high) {
    return (unsigned long long)(unsigned)low | ((unsigned long long)(unsigned)(high + (low >> 31)) << 32);
}
__device__ __forceinline__ void unpack_pair(unsigned long long w, int& low, int& high) {
    low = (int)(unsigned)w;
    high = (int)(unsigned)(w >> 32) - (low >> 31);
}

// Maximum of a non-negative float over the wave, in the DPP network (no LDS round trips: __shfl_xor is ds_bpermute, six
// dependent ~60-cycle LDS operations per value).  Non-negative floats order like their bit patterns, so the maximum is
// taken on integers (no NaN canonicalisation instructions; a NaN bound comes out as the maximum, which scale_for maps to 1).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umax(unsigned v) {
    // lanes the pattern does not feed (masked rows) read their own value: old = v, bound_ctrl off
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
    return max(v, o);
}
// Sum over the wave in the same network (fixed tree: reproducible).  Lanes a masked row pattern does not feed add 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(o);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1, 0xf>(v);
    v = dpp_add<0x4E, 0xf>(v);
    v = dpp_add<0x141, 0xf>(v);
    v = dpp_add<0x140, 0xf>(v);     // every lane of a row of 16 holds the row sum
    v = dpp_add<0x142, 0xa>(v);     // rows 1, 3 += rows 0, 2
    v = dpp_add<0x143, 0xc>(v);     // rows 2, 3 += row 1 (= rows 0 + 1): lane 63 holds the wave sum
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_nonneg(float x) {
    unsigned v = __float_as_uint(x);
    v = dpp_umax<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
    v = dpp_umax<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
    v = dpp_umax<0x141, 0xf>(v);    // row_half_mirror
    v = dpp_umax<0x140, 0xf>(v);    // row_mirror: every lane of a row of 16 holds the row maximum
    v = dpp_umax<0x142, 0xa>(v);    // row_bcast:15 into rows 1 and 3
    v = dpp_umax<0x143, 0xc>(v);    // row_bcast:31 into rows 2 and 3: lane 63 holds the wave maximum
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)v, 63));
}

// Phase trace (pixie_mpm_set_scalar "trace" 1; read back with pixie::mpm_trace_read, not part of the C ABI): per work item 8
// 100 MHz timestamps -- start, tile staged, particles updated (G2P + stress), scales known, scatter done, tile published.
// The trace buffer, the F_TRACE instantiation and every diagnostic entry point exist only in the -DPIXIE_DIAG build
// (libpixie_hip_diag.so: tests and profilers); the production library carries none of it.
#ifdef PIXIE_DIAG
constexpr int kMpmTraceItems = 32768;
__device__ unsigned long long g_mpm_trace[kMpmTraceItems * 8];
#define PX_MPM_STAMP(i) do { if ((FL & F_TRACE) && (sp.trace & 1) && tid == 0 && blockIdx.x < (unsigned)kMpmTraceItems) g_mpm_trace[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define PX_MPM_STAMP(i) do { } while (0)
#endif

// kernel variants (FL)
constexpr int F_TRACE = 1;    // phase stamps + the ablation switches of StepParams.trace (timing studies only)
constexpr int F_PACK32 = 2;   // packed 32-bit scatter (above)
constexpr int F_WIDE = 8;     // no scheduling barriers: for scenes too small to fill the chip, where latency, not issue, binds

// OCC = waves per SIMD the register allocation is held to (launch_bounds).  Built without the SLP vectoriser (see
// pixie_amd/build.py) the kernel needs 96 VGPRs -> 5 waves per SIMD with no spills (with it: 168 VGPRs, 3 waves, and
// 20 % slower); 6 -> 80 VGPRs with ~20 spilled dwords (measured slower: 86 vs 81 us at 1 M particles).  Chosen at run
// time (set_scalar "occupancy"), same arithmetic.
template <bool DO_G2P, bool DO_P2G, int OCC, int FL>
__global__ __launch_bounds__(kWG, OCC) void mpm_block_kernel(MpmPtrs S, StepParams sp, PModSet pms) {
    constexpr bool PACK = (FL & F_PACK32) != 0;
    constexpr bool TRACE = (FL & F_TRACE) != 0;
    constexpr bool SCHED = (FL & F_WIDE) == 0;
    __shared__ float4 tv[kTN];    // grid velocities of the tile (G2P source)
    __shared__ unsigned long long ta[PACK ? 2 : 4][kTN];  // (m*v.xyz, m) of this work item as scaled integers (P2G target)
    __shared__ float s_red[2][kWG / 64];
    // Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8) and each XCD has its own L2.  The work list is in
    // block order, so with sp.xcd_order every XCD takes a CONTIGUOUS eighth of it: neighbouring blocks -- whose 8^3 tiles of gout
    // overlap eightfold -- then share their staging reads in that XCD's L2 instead of fetching them once per XCD
    // (set_scalar "xcd_order", on by default: -2.6 % per substep at 1 M and 100 k, bit-identical results; the tile a work item
    // publishes is addressed by the item, not by the workgroup).
    int item = (int)blockIdx.x;
    if (sp.xcd_order) {
        const int per = (int)gridDim.x >> 3;
        if (item < (per << 3)) item = (item & 7) * per + (item >> 3);      // (the last n % 8 items keep their place)
    }
    const int4 it = S.items[item];
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;   // = the work-item capacity of the current binning (256; 128 on request)
    const int bz = it.x % S.nbk, by = (it.x / S.nbk) % S.nbk, bx = it.x / (S.nbk * S.nbk);
    const int ox = bx * kBS - 1, oy = by * kBS - 1, oz = bz * kBS - 1;
    const int ng = S.ng;
    PX_MPM_STAMP(0);
    Preload L;
    L.selection = 1;
    if (tid < it.z) preload_particle<DO_G2P, DO_P2G>(S, it.y + tid, L);   // in flight while the tile is staged
    for (int idx = tid; idx < kTN; idx += nthr) {
        if (DO_G2P) {
            const int gz = oz + (idx & (kTS - 1)), gy = oy + ((idx >> 3) & (kTS - 1)), gx = ox + (idx >> 6);
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)gx < (unsigned)ng && (unsigned)gy < (unsigned)ng && (unsigned)gz < (unsigned)ng && !(TRACE && (sp.trace & 0x400)))
                g = S.gout[((size_t)gx * ng + gy) * ng + gz];
            tv[idx] = g;
        }
        if (DO_P2G) {
            ta[0][idx] = 0ull; ta[1][idx] = 0ull;
            if (!PACK) { ta[2][idx] = 0ull; ta[3][idx] = 0ull; }
        }
    }
    __syncthreads();
    PX_MPM_STAMP(1);

    // One chunk of <= 256 particles per work item.  Sharing one tile between more particles was measured both ways and
    // loses: (a) a workgroup looping over several 256-particle chunks (integer sums folded into an fp32 tile between
    // chunks) -- hipcc 7.2 keeps 176 VGPRs live across the loop (3 waves per SIMD instead of 5); (b) work items of 384 ...
    // 1024 threads -- 107 ... 132 us per launch at 1 M particles against 81 us for 256 (r2g): every barrier then waits
    // for the slowest of 6 ... 16 waves.  The per-item costs (staging, zeroing, publish) are the smaller evil.
    const int q = tid;
    ScatterIn in;
    in.active = false;
    if (q < it.z) particle_phase1<DO_G2P, DO_P2G, SCHED>(S, sp, pms, it.y + q, ox, oy, oz, tv, L, in);
    if (!DO_P2G) return;
    PX_MPM_STAMP(2);

    // ---- P2G: a particle whose stencil left the tile goes straight to HBM (fp32 atomics into gin) ----
    Stencil st;
    int b0 = -1;
    if (in.active) {
        st = make_stencil(in.x[0], in.x[1], in.x[2], S.inv_dx);
        if (!stencil_inside(st, ng)) {
            atomicAdd(S.oob, 1ull);
            S.selection[it.y + q] = 2;
            Mat3 Fnow;
#pragma unroll
            for (int i = 0; i < 9; ++i) Fnow.m[i] = S.F[i * S.n + it.y + q];     // (the return-mapped F this launch has just stored)
            freeze_particle_state(S, it.y + q, Fnow);
            in.active = false;
        } else {
            const int lx = st.base[0] - ox, ly = st.base[1] - oy, lz = st.base[2] - oz;
            if ((unsigned)lx <= (unsigned)(kTS - 3) && (unsigned)ly <= (unsigned)(kTS - 3) && (unsigned)lz <= (unsigned)(kTS - 3)) {
                b0 = (lx * kTS + ly) * kTS + lz;
            } else {
                atomicAdd(S.oob + 1, 1ull);
                float mvAT[21];
#pragma unroll
                for (int a = 0; a < 3; ++a) mvAT[a] = in.mv[a];
#pragma unroll
                for (int k = 0; k < 9; ++k) { mvAT[3 + k] = in.A.m[k]; mvAT[12 + k] = in.T.m[k]; }
                if (!p2g_scatter_global(S.gin, S.blk_flags, S.nbk, ng, st, mvAT, in.mass)) atomicAdd(S.oob + 2, 1ull);
                in.active = false;
            }
        }
    }
    // ---- workgroup bounds -> power-of-two scales ----
    // One contribution of a particle is  w (mv_a + A_a . d) + T_a . g  with  w <= 0.75^3, |d_b| <= 1.5, |g_b| <= 0.75^2 (the
    // B-spline weights and their derivatives in cell units), so  r_p = max_a [0.421875 (|mv_a| + 1.5 sum_b |A_ab|) +
    // 0.5625 sum_b |T_ab|]  bounds every contribution of particle p.
    //   exact mode: scale by the workgroup MAXIMUM of r_p to [2^41, 2^42): 256 contributions stay below 2^50.
    //   packed mode: scale by the workgroup SUM of r_p to [2^29, 2^30): |any node sum| <= sum_p r_p < 2^30 whatever the
    //     particle count, and the quantum is 2^-30 of the SUM instead of 2^-22 of 256 maxima -- typically 10-20x finer
    //     (the sum of ~180 bounds of which most are well below the largest).  Same for the masses.
    float bp = 0.0f, bm = 0.0f;
    if (TRACE && (sp.trace & 0x200)) { bp = PACK ? 256.0f : 1.0f; bm = PACK ? 0.256f : 1e-3f; }
    else {
        if (in.active) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float r = 0.421875f * (fabsf(in.mv[a]) + 1.5f * (fabsf(in.A.m[3 * a]) + fabsf(in.A.m[3 * a + 1]) + fabsf(in.A.m[3 * a + 2]))) +
                                0.5625f * (fabsf(in.T.m[3 * a]) + fabsf(in.T.m[3 * a + 1]) + fabsf(in.T.m[3 * a + 2]));
                bp = fmaxf(bp, r);
            }
            bm = 0.421875f * in.mass;
        }
        if (PACK) { bp = wave_sum(bp); bm = wave_sum(bm); }
        else { bp = wave_max_nonneg(bp); bm = wave_max_nonneg(bm); }
        if ((tid & 63) == 0) { s_red[0][tid >> 6] = bp; s_red[1][tid >> 6] = bm; }
        __syncthreads();
        bp = s_red[0][0]; bm = s_red[1][0];
        for (int w = 1; w < (nthr >> 6); ++w) {
            if (PACK) { bp += s_red[0][w]; bm += s_red[1][w]; }     // (fixed order: the scale is reproducible)
            else { bp = fmaxf(bp, s_red[0][w]); bm = fmaxf(bm, s_red[1][w]); }
        }
    }
    constexpr int kTop = PACK ? 29 : 41;
    const float sP = scale_for(bp, kTop), sM = scale_for(bm, kTop);
    PX_MPM_STAMP(3);

    if (in.active) {
#pragma unroll
        for (int a = 0; a < 3; ++a) in.mv[a] *= sP;
#pragma unroll
        for (int k = 0; k < 9; ++k) { in.A.m[k] *= sP; in.T.m[k] *= sP; }
        p2g_scatter<SCHED>(st, in.mv, in.A, in.T, in.mass * sM, [&](int i, int j, int k, const float mom[3], float m) {
            const int idx = b0 + (i * kTS + j) * kTS + k;
            if (TRACE && (sp.trace & 0x100)) { asm volatile("" :: "v"(mom[0]), "v"(mom[1]), "v"(mom[2]), "v"(m)); return; }
            if (PACK) {
                // v_cvt_rpi rounds exact ties UP, and ties are common (a contribution of magnitude 2^22 is a float with one
                // fractional bit): left alone that is a drift of ~0.2 quanta per contribution in +x, +y, +z -- measured as
                // 3e-3 of the total momentum over 500 substeps.  Neighbouring nodes therefore alternate: (i + j + k) even
                // adds round(x), odd SUBTRACTS round(-x), i.e. rounds ties down.  Unbiased, and still a pure function of
                // the particle's own data (deterministic, order-independent).
                if (((i + j + k) & 1) == 0) {
                    atomicAdd(&ta[0][idx], pack_pair(round_to_int(mom[0]), round_to_int(mom[1])));
                    // the mass is never negative: in the low half it needs no borrow correction
                    atomicAdd(&ta[1][idx], (unsigned long long)(unsigned)round_to_int(m) | ((unsigned long long)(unsigned)round_to_int(mom[2]) << 32));
                } else {
                    __hip_atomic_fetch_sub(&ta[0][idx], pack_pair(round_to_int_neg(mom[0]), round_to_int_neg(mom[1])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_sub(&ta[1][idx], pack_pair(round_to_int_neg(m), round_to_int_neg(mom[2])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                atomicAdd(&ta[0][idx], to_fixed(mom[0]));
                atomicAdd(&ta[1][idx], to_fixed(mom[1]));
                atomicAdd(&ta[2][idx], to_fixed(mom[2]));
                atomicAdd(&ta[3][idx], to_fixed(m));
            }
        });
    }
    // where this thread's nodes go in the staged tile: fetched now, consumed behind the barrier
    const unsigned lut = (nthr == kWG) ? S.staged_lut[tid] : 0u;
    __syncthreads();
    PX_MPM_STAMP(4);
    const float iP = pow2_reciprocal(sP), iM = pow2_reciprocal(sM);
    // ---- publish the tile: coalesced stores; the grid update sums the tiles that cover each node ----
    float4* dst = S.part + (size_t)item * kTN;
    if (!(TRACE && (sp.trace & 0x800)))
        for (int idx = tid; idx < kTN; idx += nthr) {
            float4 o;
            if (PACK) {
                int px, py, pm, pz;
                unpack_pair(ta[0][idx], px, py);
                const unsigned long long w1 = ta[1][idx];
                pm = (int)(unsigned)w1; pz = (int)(unsigned)(w1 >> 32);
                o = make_float4((float)px * iP, (float)py * iP, (float)pz * iP, (float)pm * iM);
            } else {
                o = make_float4(from_fixed(ta[0][idx], iP), from_fixed(ta[1][idx], iP), from_fixed(ta[2][idx], iP), from_fixed(ta[3][idx], iM));
            }
            // (staged_index is ~30 instructions of selects per node; with the usual 256-thread work items each thread's two
            // nodes are tid and tid + 256 and their staged positions come from a 1 KB table, two 16-bit halves of one word)
            const int si = (nthr == kWG) ? (int)((idx < kWG) ? (lut & 0xffffu) : (lut >> 16)) : staged_index(idx >> 6, (idx >> 3) & 7, idx & 7);
            // Typically 40-60 % of a tile's nodes received nothing (the drift margin planes, corners beyond every stencil): they
            // are neither stored nor -- by the mask -- read back.  Adding an all-zero float4 is a no-op, so the sums are unchanged.
            const bool nz = (o.x != 0.0f) | (o.y != 0.0f) | (o.z != 0.0f) | (o.w != 0.0f);
            if (S.sparse_tiles) {
                const unsigned long long live = __ballot(nz);     // lanes of a wave hold 64 consecutive nodes
                if ((tid & 63) == 0) S.tile_mask[(size_t)item * 8 + (idx >> 6)] = live;
                if (nz) dst[si] = o;
            } else {
                dst[si] = o;
            }
        }
    PX_MPM_STAMP(5);
#ifdef PIXIE_DIAG
    if (TRACE && (sp.trace & 1) && tid == 0 && blockIdx.x < (unsigned)kMpmTraceItems)   // where it ran: HW_ID | XCC_ID << 32
        g_mpm_trace[blockIdx.x * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
#endif
}

// ------------------------------------------------------------------ re-binning (counting sort by block)
__device__ __forceinline__ int block_of(const MpmPtrs& S, int p) {
    int b[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int base = (int)(S.x[d * S.n + p] * S.inv_dx - 0.5f);
        base = max(0, min(base, S.ng - 3));
        b[d] = base / kBS;
    }
    return (b[0] * S.nbk + b[1]) * S.nbk + b[2];
}

// key[p] = block of particle p; rank[p] = its arrival number inside the block.  Lanes of a wave that share a key
// (the common case once the particles are binned) issue one atomic for the whole group.
__global__ __launch_bounds__(256) void bin_count_kernel(MpmPtrs S, int* __restrict__ keys, int* __restrict__ rank,
                                                        int* __restrict__ counts, unsigned* __restrict__ drift2_bits) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < S.n;
    const int key = valid ? block_of(S, p) : -1;
    {   // largest squared displacement since the last re-binning (drives the cadence, see rebin())
        float d2 = 0.0f;
        if (valid) {
            for (int d = 0; d < 3; ++d) { const float e = S.x[d * S.n + p] - S.xref[d * S.n + p]; d2 += e * e; }
            if (!(d2 < 3.0e38f)) d2 = 3.0e38f;
        }
        for (int off = 32; off > 0; off >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, off, 64));
        if ((threadIdx.x & 63) == 0 && d2 > __uint_as_float(*drift2_bits)) atomicMax(drift2_bits, __float_as_uint(d2));
    }
    // Two groups of a wave share one atomic each -- the key of its first lane and the key of the first lane outside that group (binned
    // particles arrive block by block: a wave rarely spans more) -- and whoever is left takes its own.  All of a wave's atomics are
    // then in flight within three round trips; until round 6 every distinct key of the wave cost a dependent one (a scene in motion:
    // 4-6 per wave, 99 us for 1 M particles).  The arrival number is arbitrary either way; bin_local_order_kernel makes the order a
    // function of the particle data.
    const int lane = threadIdx.x & 63;
    int my_rank = 0;
    unsigned long long todo = __ballot(valid);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (!todo) break;
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader);
        const unsigned long long grp = __ballot(valid && key == k) & todo;
        int base = 0;
        if (lane == leader) base = atomicAdd(&counts[k], __popcll(grp));
        base = __shfl(base, leader);
        if ((grp >> lane) & 1ull) my_rank = base + __popcll(grp & ((1ull << lane) - 1ull));
        todo &= ~grp;
    }
    if ((todo >> lane) & 1ull) my_rank = atomicAdd(&counts[key], 1);
    if (valid) { keys[p] = key; rank[p] = my_rank; }
}

// Exclusive scan of the block counts + the work list (<= cap particles per item), in three launches over chunks of 1024 blocks:
// chunk totals -> scan of the chunk totals (one workgroup) -> scan inside each chunk + the writes.  (Until round 5 ONE 1024-thread
// workgroup walked all blocks: 58 us for the 27 000 blocks of a 120^3 grid, 280 us for the 125 000 of the reference's sand configuration
// at n_grid 200 -- 7 % of that scene's run at a re-binning every ~33 substeps; profiles/r5last_stats_sand.csv.)
constexpr int kScanChunk = 1024;
struct ScanTriple { int cnt, itm, other; };   // particles, work items at `cap`, work items at the other capacity (auto item_cap, see rebin())
__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}
// exclusive prefix of v over the 1024-thread workgroup; *total = the workgroup's sum
__device__ __forceinline__ int wg_excl_scan_1024(int v, int* s_wave /* [16] */, int* total) {
    const int incl = wave_incl_scan(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 63) s_wave[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int t = s_wave[k]; if (k < w) base += t; tot += t; }
    *total = tot;
    return base + incl - v;
}
__device__ __forceinline__ ScanTriple scan_triple_of(int c, int cap) {
    const int other = (cap == kWG) ? kWG / 2 : kWG;
    return ScanTriple{c, (c + cap - 1) / cap, (c + other - 1) / other};
}
__global__ __launch_bounds__(kScanChunk) void bin_scan_partial_kernel(const int* __restrict__ counts, ScanTriple* __restrict__ chunk, int nblocks, int cap) {
    __shared__ int s_w[3][16];
    const int b = blockIdx.x * kScanChunk + threadIdx.x;
    const ScanTriple t = scan_triple_of(b < nblocks ? counts[b] : 0, cap);
    int v[3] = {t.cnt, t.itm, t.other};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
        if ((threadIdx.x & 63) == 0) s_w[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ScanTriple r{0, 0, 0};
        for (int w = 0; w < 16; ++w) { r.cnt += s_w[0][w]; r.itm += s_w[1][w]; r.other += s_w[2][w]; }
        chunk[blockIdx.x] = r;
    }
}
// chunk totals -> exclusive prefixes (in place); n_items[0] = work items, n_items[3] = what the other capacity would make.  One workgroup.
__global__ __launch_bounds__(1024) void bin_scan_chunks_kernel(ScanTriple* __restrict__ chunk, int nchunks, int* __restrict__ n_items) {
    __shared__ int s_w[16];
    const int per = (nchunks + 1023) / 1024;
    const int c0 = min((int)threadIdx.x * per, nchunks), c1 = min(c0 + per, nchunks);
    ScanTriple mine{0, 0, 0};
    for (int c = c0; c < c1; ++c) { mine.cnt += chunk[c].cnt; mine.itm += chunk[c].itm; mine.other += chunk[c].other; }
    int tc, ti, to;
    int pc = wg_excl_scan_1024(mine.cnt, s_w, &tc);
    int pi = wg_excl_scan_1024(mine.itm, s_w, &ti);
    int po = wg_excl_scan_1024(mine.other, s_w, &to);
    for (int c = c0; c < c1; ++c) {
        const ScanTriple t = chunk[c];
        chunk[c] = ScanTriple{pc, pi, po};
        pc += t.cnt; pi += t.itm; po += t.other;
    }
    if (threadIdx.x == 0) { n_items[0] = ti; n_items[3] = to; }
}
__global__ __launch_bounds__(kScanChunk) void bin_scan_write_kernel(const int* __restrict__ counts, const ScanTriple* __restrict__ chunk,
                                                                    int* __restrict__ offsets, int4* __restrict__ items,
                                                                    int2* __restrict__ blk_items, int nblocks, int cap) {
    __shared__ int s_w[16];
    const int b = blockIdx.x * kScanChunk + threadIdx.x;
    const int cnt = b < nblocks ? counts[b] : 0;
    const int nit = (cnt + cap - 1) / cap;
    const ScanTriple base = chunk[blockIdx.x];
    int dummy;
    const int c = base.cnt + wg_excl_scan_1024(cnt, s_w, &dummy);
    int i = base.itm + wg_excl_scan_1024(nit, s_w, &dummy);
    if (b >= nblocks) return;
    offsets[b] = c;
    blk_items[b] = make_int2(i, nit);
    for (int j = 0; j < cnt; j += cap) items[i++] = make_int4(b, c + j, min(cap, cnt - j), 0);
}

__global__ __launch_bounds__(256) void bin_order_kernel(const int* __restrict__ keys, const int* __restrict__ rank,
                                                        const int* __restrict__ offsets, int* __restrict__ order, int n) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < n) order[offsets[keys[p]] + rank[p]] = p;
}

// Order inside a block: round-robin over its 64 cells -- position = (rank within the cell, cell).  Consecutive lanes of
// the block kernel then sit in consecutive cells, so the 27 same-offset LDS accesses of a wave (G2P reads and P2G
// ds_add_u64) hit distinct nodes in distinct banks instead of colliding 3-4 ways as they do in arrival order.
// One workgroup per block; `order` holds the block-contiguous listing, `order2` receives the final one.
//
// The rank within the cell is the particle's rank by its CURRENT SLOT among the block's particles of that cell -- not
// its arrival number (`order` is filled through atomics, so arrival differs from run to run).  That makes the listing,
// and with it the split of a crowded block into 256-particle work items and their fixed-point scales, a pure function
// of the particle data: together with the integer tile sums and the fixed summation order of the grid kernel, a rollout
// is bit-reproducible from run to run (as long as no particle takes the slow path's fp32 atomics).  Counting
// "same cell, lower slot" is quadratic in the block's population -- cnt^2 / 256 compares per thread through LDS tiles,
// ~20 us for the 800-particle blocks of the 100 k scene, once per re-binning.
__device__ __forceinline__ int cell_in_block(const MpmPtrs& S, int p) {
    int c = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int base = (int)(S.x[d * S.n + p] * S.inv_dx - 0.5f);
        base = max(0, min(base, S.ng - 3));
        c = c * kBS + (base & (kBS - 1));
    }
    return c;
}
// Material class of a particle for the ordering inside a block: the constitutive branches of return_map_and_stress.  A block
// that mixes materials (material_mode=neural uploads an id per particle, PG/material_field.py:343-363) is laid out class by
// class, round-robin by cell inside a class, so that a wave runs ONE branch instead of all of them (mixed 1 M scene: 1955 -> 1600
// VALU instructions per wave, 82.9 -> 78.5 us per substep = the single-material plastic scenes' 78.2; profiles/r5d_*); a
// single-material block -- every block of a single-material scene -- keeps the order it always had.
constexpr int kMatClasses = 7;
__device__ __forceinline__ int material_class(int material) {
    return material == 0 ? 0 : material == 1 ? 1 : material == 2 ? 2 : material == 3 ? 3 : material == 5 ? 4 : material == 6 ? 5 : 6;
}
__global__ __launch_bounds__(256) void bin_local_order_kernel(MpmPtrs S, const int* __restrict__ counts, const int* __restrict__ offsets,
                                                              const int* __restrict__ order, int* __restrict__ cellk,
                                                              int* __restrict__ rank, int* __restrict__ order2) {
    constexpr int kCells = kBS * kBS * kBS;
    __shared__ int cc[kMatClasses * kCells];     // particles per (class, cell)
    __shared__ int s_class_base[kMatClasses];
    __shared__ int2 s_tile[256];
    const int b = blockIdx.x;
    const int cnt = counts[b];
    if (cnt == 0) return;
    const int off = offsets[b];
    for (int t = threadIdx.x; t < kMatClasses * kCells; t += 256) cc[t] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
        const int p = order[off + t];
        const int c = material_class(S.material[p]) * kCells + cell_in_block(S, p);   // key: (class, cell)
        cellk[off + t] = c;
        atomicAdd(&cc[c], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int m = 0; m < kMatClasses; ++m) {
            s_class_base[m] = acc;
            for (int k = 0; k < kCells; ++k) acc += cc[m * kCells + k];
        }
    }
    __syncthreads();
    for (int t0 = 0; t0 < cnt; t0 += 256) {
        const int t = t0 + threadIdx.x;
        const bool mine = t < cnt;
        const int c = mine ? cellk[off + t] : -1, p = mine ? order[off + t] : 0;
        int r = 0;
        for (int u0 = 0; u0 < cnt; u0 += 256) {
            __syncthreads();
            const int u = u0 + threadIdx.x;
            s_tile[threadIdx.x] = (u < cnt) ? make_int2(cellk[off + u], order[off + u]) : make_int2(-2, 0);
            __syncthreads();
            const int m = min(256, cnt - u0);
            for (int k = 0; k < m; ++k) {
                const int2 e = s_tile[k];
                r += (e.x == c && e.y < p) ? 1 : 0;
            }
        }
        if (mine) {      // slot = particles of earlier classes + (round r, cell) position inside the class
            const int m = c / kCells, cl = c - m * kCells;
            int pos = s_class_base[m];
            for (int k = 0; k < kCells; ++k) {
                const int n = cc[m * kCells + k];
                pos += min(n, r) + ((k < cl && n > r) ? 1 : 0);
            }
            order2[off + pos] = p;
        }
    }
}

// blk_flags bit 0 <- "one of my 27 neighbours holds particles", and the compact list of those ACTIVE blocks: the only
// ones the grid kernel is launched for (workgroup dispatch alone costs ~10 us for the 27000 blocks of a 120^3 grid).
__global__ __launch_bounds__(256) void bin_mark_active_kernel(const int* __restrict__ counts, int* __restrict__ blk_flags,
                                                              int* __restrict__ active_list, int* __restrict__ n_active, int nbk,
                                                              const int2* __restrict__ blk_items, int2* __restrict__ nbr_table) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nbk * nbk * nbk) return;
    const int bz = b % nbk, by = (b / nbk) % nbk, bx = b / (nbk * nbk);
    bool active = false;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const int x = bx + dx, y = by + dy, z = bz + dz;
                if ((unsigned)x < (unsigned)nbk && (unsigned)y < (unsigned)nbk && (unsigned)z < (unsigned)nbk)
                    active = active || counts[(x * nbk + y) * nbk + z] > 0;
            }
    blk_flags[b] = active ? 1 : 0;
    if (!active) return;
    const int slot = atomicAdd(n_active, 1);
    active_list[slot] = b;
    // everything the grid kernel needs to find this block's tiles, in one 224-byte row: one memory round trip there
    // instead of active_list -> blk_items -> tiles
    int2* row = nbr_table + (size_t)slot * 28;
    row[0] = make_int2(b, 0);
    for (int q = 0; q < 27; ++q) {
        const int x = bx + q / 9 - 1, y = by + (q / 3) % 3 - 1, z = bz + q % 3 - 1;
        const bool in = (unsigned)x < (unsigned)nbk && (unsigned)y < (unsigned)nbk && (unsigned)z < (unsigned)nbk;
        row[1 + q] = in ? blk_items[(x * nbk + y) * nbk + z] : make_int2(0, 0);
    }
}

// dst[r][q] = src[r][order[q]] for every row of the particle word array
// `skip_vcft`: the launch this re-binning precedes starts with a G2P, which overwrites v, C and F_trial of every particle that takes
// part (in the fused kernel they are not even stored between the gather and the scatter) -- 21 of the 51 rows are dead for those
// particles and only travel for the ones that sit out (selection != 0: deselected by the caller or frozen outside the grid), whose
// last stored values a caller may still read.
__global__ __launch_bounds__(256) void bin_permute_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst,
                                                          const int* __restrict__ order, int n, int rows_per_y, int skip_vcft) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const int p = order[q];
    const int r0 = blockIdx.y * rows_per_y, r1 = min(r0 + rows_per_y, (int)R_COUNT);
    const bool sits_out = skip_vcft && src[(size_t)R_SELECTION * n + p] != 0u;
    for (int r = r0; r < r1; ++r) {
        const bool dead = skip_vcft && ((r >= R_V && r < R_V + 3) || (r >= R_FT && r < R_FT + 9) || (r >= R_C && r < R_C + 9));
        if (dead && !sits_out) continue;
        const int rs = (r >= R_XREF) ? (R_X + r - R_XREF) : r;   // xref <- x: positions at this re-binning
        dst[(size_t)r * n + q] = src[(size_t)rs * n + p];
    }
}

__global__ void iota_kernel(int* dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = i;
}

// ------------------------------------------------------------------ grid kernel
// grid_normalization_and_gravity (mpm_utils.py:398-409), add_damping_via_grid (:583-588) and the
// BC `collide` closures (mpm_solver_warp.py:785-840 surface, :874-897 cuboid, :917-974 bounding box)
// in one sweep; also re-zeroes (m*v, m) so the next P2G starts from a clean grid (zero_grid :295-300).
// (No FMA contraction in apply_bc / finish_node: whether `float(k) * dx - point` is fused decides on which side of a collider plane a
// node plane lies (tests: knife-edge scene), and hipcc's choice depends on the code around it.  Every operation rounded, as in the
// float32 oracle -- the same bits in whatever kernel these are inlined into.)
__device__ __forceinline__ void apply_bc(const BCDev& b, int ix, int iy, int iz, int ng, float dx, float time,
                                         float dt, float v[3]) {
#pragma clang fp contract(off)
    if (b.type == PIXIE_BC_SURFACE) {
        if (time >= b.start && time < b.end) {
            const float ox = (float)ix * dx - b.point[0], oy = (float)iy * dx - b.point[1], oz = (float)iz * dx - b.point[2];
            const float dp = ox * b.normal[0] + oy * b.normal[1] + oz * b.normal[2];
            if (dp < 0.0f) {
                if (b.surface_type == 11) {
                    if ((float)iz * dx < 0.4f || (float)iz * dx > 0.53f) {
                        v[0] = v[1] = v[2] = 0.0f;
                    } else {
                        v[0] = v[0] * 0.3f; v[1] = 0.0f; v[2] = v[2] * 0.3f;
                    }
                } else {
                    // sticky, and -- as in the reference (:821-840) -- slip/friction too: the projected
                    // velocity is computed there but the store is zero.
                    v[0] = v[1] = v[2] = 0.0f;
                }
            }
        }
    } else if (b.type == PIXIE_BC_CUBOID) {
        if (time >= b.start && time < b.end) {
            const float ox = (float)ix * dx - b.point[0], oy = (float)iy * dx - b.point[1], oz = (float)iz * dx - b.point[2];
            if (fabsf(ox) < b.size[0] && fabsf(oy) < b.size[1] && fabsf(oz) < b.size[2]) {
                v[0] = b.velocity[0]; v[1] = b.velocity[1]; v[2] = b.velocity[2];
            }
        } else if (b.reset == 1) {
            if (time < b.end + 15.0f * dt) v[0] = v[1] = v[2] = 0.0f;
        }
    } else {
        const int padding = 3;
        if (time >= b.start && time < b.end) {
            if (ix < padding && v[0] < 0.0f) v[0] = 0.0f;
            if (ix >= ng - padding && v[0] > 0.0f) v[0] = 0.0f;
            if (iy < padding && v[1] < 0.0f) v[1] = 0.0f;
            if (iy >= ng - padding && v[1] > 0.0f) v[1] = 0.0f;
            if (iz < padding && v[2] < 0.0f) v[2] = 0.0f;
            if (iz >= ng - padding && v[2] > 0.0f) v[2] = 0.0f;
        }
    }
}

// (m*v, m) of node (gx,gy,gz) = what slow-path particles added to gin + the tiles of the work items that cover the node.
// A node with g = 4m + r along an axis lies in the tiles of blocks {m-1, m} (r < 3) or {m, m+1} (r = 3) on that axis:
// 8 candidate blocks per node, all among the 27 neighbours of the node's own block.  The wave first fetches the 27
// (first item, count) pairs with one load (lane i < 27), then every lane walks its 8 candidates in rounds so that the
// 8 tile loads of a round are in flight together.  The order of the sum is fixed; every staged value is read once.
// lane i < 27 fetches (first item, count) of neighbour block i of block (Bx,By,Bz); other lanes get (0,0)
__device__ __forceinline__ int2 neighbour_items(const MpmPtrs& S, int Bx, int By, int Bz) {
    const int lane = threadIdx.x & 63;
    int2 mine = make_int2(0, 0);
    if (lane < 27) {
        const int bx = Bx + lane / 9 - 1, by = By + (lane / 3) % 3 - 1, bz = Bz + lane % 3 - 1;
        if ((unsigned)bx < (unsigned)S.nbk && (unsigned)by < (unsigned)S.nbk && (unsigned)bz < (unsigned)S.nbk)
            mine = S.blk_items[(bx * S.nbk + by) * S.nbk + bz];
    }
    return mine;
}
// One axis of staged_index: tile coordinate t -> (first node of its sub-box, nodes in the sub-box, t - first)
__device__ __forceinline__ void staged_axis(int t, int& o, int& n, int& r) {
    o = (t == 0) ? 0 : (t <= 4 ? 1 : 5);
    n = (t == 0) ? 1 : (t <= 4 ? 4 : 3);
    r = t - o;
}
// RB = items per candidate block fetched in one go (8 x RB tile loads in flight).
//
// Round 6: this gather was written as if the grid kernel were latency-bound, and its counters say otherwise -- 816 VALU instructions per
// wave for ~10 float4 additions per lane: 3.7 waves per SIMD x 1.5 us of VALU issue at 1 M, FIFTEEN waves per SIMD on the reference's sand
// configuration (15 700 active blocks: 21 of the launch's 21 us).  What it spent them on: staged_index per candidate (~30 selects x 8), a
// branch + four zeroing moves around every predicated load, 64-bit address arithmetic, a 64-bit `live` word assembled bit by bit.  Now:
// staged_index taken apart by axis (six small selects per axis and side, ~6 multiply-adds per candidate); 32-bit element offsets from
// uniform bases; and NO predication -- a load that has nothing to fetch reads the all-zero tile at S.zero_off (or mask word 0, whose
// value is then ignored), so every lane issues the same 8 x RB loads and adds what comes back.  Adding +0 changes nothing and the order of
// the sum is still (round, candidate) whatever RB: every instantiation returns the bits it returned before.
// (Round 4 tried issuing the first round's tile loads TOGETHER with the mask loads and dropping unstored nodes afterwards --
// one dependent round trip fewer: 12.7 -> 15.5 us per launch at 1 M, the extra requests cost more than the trip saves;
// profiles/r4f_mpm_grid_speculative_tile_loads_rejected.txt.  The loads added here all hit ONE line.)
template <int RB>
__device__ __forceinline__ float4 gather_node(const MpmPtrs& S, int2 mine, int lx, int ly, int lz, float4 acc) {
    const int ax = (lx == 3) ? 0 : -1, ay = (ly == 3) ? 0 : -1, az = (lz == 3) ? 0 : -1;  // first candidate offset per axis
    // this node inside the tiles of the two candidate blocks per axis: coordinates t0 and t0 - 4
    const int t0x = lx - 4 * ax + 1, t0y = ly - 4 * ay + 1, t0z = lz - 4 * az + 1;
    int ox[2], nx[2], rx[2], oy[2], ny[2], ry[2], oz[2], nz[2], rz[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        staged_axis(t0x - 4 * b, ox[b], nx[b], rx[b]);
        staged_axis(t0y - 4 * b, oy[b], ny[b], ry[b]);
        staged_axis(t0z - 4 * b, oz[b], nz[b], rz[b]);
    }
    const int src0 = (ax + 1) * 9 + (ay + 1) * 3 + (az + 1);
    unsigned toff[8];  // in float4 units from S.part: first item * kTN + staged position of this node in that block's tiles
    unsigned moff[8];  // in 32-bit words from S.tile_mask: first item * 16 + the word that holds this node's bit
    int sh[8];         // ... and the bit inside it
    int nit[8];        // items of the candidate block
    int maxc = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int bx = c >> 2, by = (c >> 1) & 1, bz = c & 1;
        const int src = src0 + bx * 9 + by * 3 + bz;
        const int first = __shfl(mine.x, src), n = __shfl(mine.y, src);
        const int bit = ((t0x - 4 * bx) * kTS + (t0y - 4 * by)) * kTS + (t0z - 4 * bz);
        const int si = ox[bx] * 64 + nx[bx] * (oy[by] * 8 + ny[by] * oz[bz]) + (rx[bx] * ny[by] + ry[by]) * nz[bz] + rz[bz];   // = staged_index(tx, ty, tz)
        toff[c] = (unsigned)first * kTN + (unsigned)si;
        moff[c] = (unsigned)first * 16u + (unsigned)(bit >> 5);
        sh[c] = bit & 31;
        nit[c] = n;
        maxc = max(maxc, n);
    }
    const unsigned* __restrict__ mask32 = reinterpret_cast<const unsigned*>(S.tile_mask);   // (little-endian halves of the 64-bit words)
    const float4* __restrict__ part = S.part;
    const unsigned zero = S.zero_off;
    for (int r0 = 0; r0 < maxc; r0 += RB) {
        unsigned idx[RB][8];
        if (S.sparse_tiles) {                // (uniform)
            unsigned w[RB][8];
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) w[r][c] = mask32[(r0 + r < nit[c]) ? moff[c] + (unsigned)(r0 + r) * 16u : 0u];
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const bool on = (r0 + r < nit[c]) && ((w[r][c] >> sh[c]) & 1u);
                    idx[r][c] = on ? toff[c] + (unsigned)(r0 + r) * kTN : zero;
                }
        } else {
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) idx[r][c] = (r0 + r < nit[c]) ? toff[c] + (unsigned)(r0 + r) * kTN : zero;
        }
        float4 q[RB][8];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) q[r][c] = part[idx[r][c]];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) { acc.x += q[r][c].x; acc.y += q[r][c].y; acc.z += q[r][c].z; acc.w += q[r][c].w; }
    }
    return acc;
}

// grid_normalization_and_gravity (mpm_utils.py:398-409), add_damping_via_grid (:583-588) and every BC for ONE node of block
// (Bx,By,Bz), from its accumulated (m*v, m); returns grid_v_out
__device__ __forceinline__ float4 finish_node(const MpmPtrs& S, const StepParams& sp, const BCSet& bcs, float4 g, int ix, int iy, int iz) {
#pragma clang fp contract(off)
    float v[3] = {0.0f, 0.0f, 0.0f};
    if (g.w > 1e-15f) {
        const float inv = 1.0f / g.w;
        v[0] = g.x * inv + sp.dt * sp.g[0];
        v[1] = g.y * inv + sp.dt * sp.g[1];
        v[2] = g.z * inv + sp.dt * sp.g[2];
    }
    if (sp.do_damping) { v[0] *= sp.damping; v[1] *= sp.damping; v[2] *= sp.damping; }
    for (int k = 0; k < bcs.n; ++k) apply_bc(bcs.bc[k], ix, iy, iz, S.ng, S.dx, sp.time, sp.dt, v);
    return make_float4(v[0], v[1], v[2], 0.0f);
}

// One WAVE updates the 4x4x4 nodes of active block `slot`: gather the staged tiles (+ what slow-path particles added to gin),
// normalise, gravity, damping, BCs -> dst.
template <int RB>
__device__ __forceinline__ void grid_block_update(const MpmPtrs& S, const StepParams& sp, const BCSet& bcs, int slot, float4* dst) {
    const int lane = threadIdx.x & 63;
#ifdef PIXIE_DIAG   // set_scalar "trace" 4: per active block four 100 MHz stamps -- start, neighbour row arrived, tiles summed, stored
#define PX_GRID_STAMP(i) do { if ((sp.trace & 4) && lane == 0 && slot < kMpmTraceItems) { asm volatile("s_waitcnt vmcnt(0)"); g_mpm_trace[slot * 8 + (i)] = wall_clock64(); } } while (0)
#else
#define PX_GRID_STAMP(i) do { } while (0)
#endif
    PX_GRID_STAMP(0);
    // one coalesced row: block id + the work items of the 27 neighbours (lane q holds neighbour q)
    const int2 row = (lane < 28) ? S.nbr_table[(size_t)slot * 28 + lane] : make_int2(0, 0);
    const int blk = __shfl(row.x, 0);
    PX_GRID_STAMP(1);
    int2 mine;
    mine.x = __shfl(row.x, (lane + 1) & 63); mine.y = __shfl(row.y, (lane + 1) & 63);
    const int Bz = blk % S.nbk, By = (blk / S.nbk) % S.nbk, Bx = blk / (S.nbk * S.nbk);
    // bit 0: active (set at re-binning); bit 1: slow-path particles added fp32 atomics into gin here
    const int flag = S.blk_flags[blk];
    const int lx = lane >> 4, ly = (lane >> 2) & 3, lz = lane & 3;
    const int ix = Bx * kBS + lx, iy = By * kBS + ly, iz = Bz * kBS + lz;
    const bool inside = ix < S.ng && iy < S.ng && iz < S.ng;
    const size_t idx = ((size_t)ix * S.ng + iy) * S.ng + iz;
    float4 g = gather_node<RB>(S, mine, lx, ly, lz, make_float4(0.f, 0.f, 0.f, 0.f));   // does not wait for the flag
    PX_GRID_STAMP(2);
    if (flag & 2) {
        if (inside) {
            const float4 q = S.gin[idx];
            g.x += q.x; g.y += q.y; g.z += q.z; g.w += q.w;
            S.gin[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        if (lane == 0) S.blk_flags[blk] = flag & 1;
    }
    if (inside) dst[idx] = finish_node(S, sp, bcs, g, ix, iy, iz);
    PX_GRID_STAMP(3);
#ifdef PIXIE_DIAG
    if ((sp.trace & 4) && lane == 0 && slot < kMpmTraceItems)
        g_mpm_trace[slot * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
#endif
}

// One wave per 4x4x4 block of nodes.  A block is ACTIVE when one of its 27 neighbours (itself included) holds particles.
// Only active blocks are read by the next G2P (a tile reaches one block beyond its own), so inactive blocks are skipped
// entirely (mode 0); their grid_v_out is brought up to date on demand (mode 1, used by the grid_v_out export) with the
// parameters of the last update, which for a massless node is just the BCs on v = 0.
// RB (tile loads in flight per candidate block) against occupancy.  The kernel is latency-bound -- a wave is three or four
// dependent memory round trips -- so what counts is how many waves are resident: a scene with several work items per block
// that fits in one round of waves (100 k particles in 50^3: 2200 active blocks) wants RB = 4 (189 VGPRs, 2 waves per SIMD); a
// scene with ~1.2 items per block and 9000 active blocks (1 M in 120^3) wants the registers back: RB = 1 = 101 VGPRs = 4 waves
// per SIMD, 14.9 -> 12.5 us per launch together with the sparse tiles.  (Holding the allocation to 80 / 64 VGPRs for 6 / 8 waves
// spills 22 / 39 dwords: 20.4 / 25.5 us.  profiles/r3s_mpm_grid_kernel_occupancy.txt)
// (Four active blocks per 256-thread workgroup instead of one wave per workgroup: measured equal at 3 700 active blocks in round 3 and again
// at the 15 700 of the reference's sand configuration in round 6 -- 21.99 vs 21.0 us, 102.0 vs 101.8 us per substep; the launch is not
// dispatch-bound.  profiles/r6c_sand_timing.txt)
template <int RB>
__global__ __launch_bounds__(64) void mpm_grid_block_kernel(MpmPtrs S, StepParams sp, BCSet bcs, int mode) {
    if (mode == 0) {
        grid_block_update<RB>(S, sp, bcs, (int)blockIdx.x, S.gout);
        return;
    }
    const int blk = (int)blockIdx.x;
    if (S.blk_flags[blk] & 1) return;
    const int Bz = blk % S.nbk, By = (blk / S.nbk) % S.nbk, Bx = blk / (S.nbk * S.nbk);
    const int lx = threadIdx.x >> 4, ly = (threadIdx.x >> 2) & 3, lz = threadIdx.x & 3;
    const int ix = Bx * kBS + lx, iy = By * kBS + ly, iz = Bz * kBS + lz;
    if (!(ix < S.ng && iy < S.ng && iz < S.ng)) return;
    S.gout[((size_t)ix * S.ng + iy) * S.ng + iz] = finish_node(S, sp, bcs, make_float4(0.f, 0.f, 0.f, 0.f), ix, iy, iz);
}

// export of grid_m / grid_v_in while a P2G is pending (tiles not yet consumed by the grid kernel)
__global__ __launch_bounds__(64) void grid_export_pending_kernel(MpmPtrs S, float* __restrict__ out, int what) {
    const int Bz = blockIdx.x % S.nbk, By = (blockIdx.x / S.nbk) % S.nbk, Bx = blockIdx.x / (S.nbk * S.nbk);
    const int lx = threadIdx.x >> 4, ly = (threadIdx.x >> 2) & 3, lz = threadIdx.x & 3;
    const int ix = Bx * kBS + lx, iy = By * kBS + ly, iz = Bz * kBS + lz;
    const bool inside = ix < S.ng && iy < S.ng && iz < S.ng;
    const size_t idx = ((size_t)ix * S.ng + iy) * S.ng + iz;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (inside) g = S.gin[idx];
    g = gather_node<4>(S, neighbour_items(S, Bx, By, Bz), lx, ly, lz, g);
    if (!inside) return;
    if (what == 0) out[idx] = g.w;
    else { out[3 * idx] = g.x; out[3 * idx + 1] = g.y; out[3 * idx + 2] = g.z; }
}

__global__ __launch_bounds__(256) void mpm_grid_kernel(MpmPtrs S, StepParams sp, BCSet bcs, int normalise) {
    const long total = (long)S.ng * S.ng * S.ng;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int iz = (int)(idx % S.ng);
    const int iy = (int)((idx / S.ng) % S.ng);
    const int ix = (int)(idx / ((long)S.ng * S.ng));
    float v[3] = {0.0f, 0.0f, 0.0f};
    if (normalise) {
        const float4 g = S.gin[idx];
        if (g.w > 1e-15f) {
            const float inv = 1.0f / g.w;
            v[0] = g.x * inv + sp.dt * sp.g[0];
            v[1] = g.y * inv + sp.dt * sp.g[1];
            v[2] = g.z * inv + sp.dt * sp.g[2];
        }
        if (sp.do_damping) { v[0] *= sp.damping; v[1] *= sp.damping; v[2] *= sp.damping; }
        if (g.x != 0.0f || g.y != 0.0f || g.z != 0.0f || g.w != 0.0f) S.gin[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    } else {
        const float4 o = S.gout[idx];
        v[0] = o.x; v[1] = o.y; v[2] = o.z;
    }
    for (int k = 0; k < bcs.n; ++k) apply_bc(bcs.bc[k], ix, iy, iz, S.ng, S.dx, sp.time, sp.dt, v);
    S.gout[idx] = make_float4(v[0], v[1], v[2], 0.0f);
}

// ------------------------------------------------------------------ set-up / utility kernels
__global__ void pmod_kernel(MpmPtrs S, StepParams sp, PModDev m) {  // overflow path (> kMaxPModFused modifiers)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S.n) return;
    float x[3], v[3];
    for (int d = 0; d < 3; ++d) { x[d] = S.x[d * S.n + p]; v[d] = S.v[d * S.n + p]; }
    apply_pmod(m, S.perm[p], sp.time, sp.dt, S.mass[p], x, v);
    for (int d = 0; d < 3; ++d) S.v[d * S.n + p] = v[d];
}

template <typename T>
__global__ void aos_to_soa_kernel(const T* __restrict__ src, T* __restrict__ dst, int n, int k, const int* perm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = perm ? perm[i] : i;
    for (int c = 0; c < k; ++c) dst[(size_t)c * n + i] = src[(size_t)s * k + c];
}
template <typename T>
__global__ void soa_to_aos_kernel(const T* __restrict__ src, T* __restrict__ dst, int n, int k, const int* perm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = perm ? perm[i] : i;
    for (int c = 0; c < k; ++c) dst[(size_t)s * k + c] = src[(size_t)c * n + i];
}
// selection 2 = "left the grid, frozen" (our marker; the reference has no such state): cleared when the caller replaces the
// positions or re-makes the grid, so that particles that are inside again take part again
__global__ void unfreeze_kernel(int* selection, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && selection[i] == 2) selection[i] = 0;
}
template <typename T>
__global__ void fill_kernel(T* dst, long count, T value) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = value;
}
__global__ void identity_F_kernel(float* Ft, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < 9; ++c) Ft[(size_t)c * n + i] = (c % 4 == 0) ? 1.0f : 0.0f;
}
// Smallest and largest positive particle mass (bit patterns of non-negative floats order like the floats).  The packed
// scatter's quantum is 2^-30 of the SUM of a work item's contribution bounds, so a node fed only by particles R times lighter
// than their neighbours in the tile is resolved to ~1e-6 R of its own mass: above a contrast of kPackMaxContrast
// the exact 64-bit mode is selected instead.
constexpr float kPackMaxContrast = 32.0f;
__global__ void mass_range_kernel(MpmPtrs S, unsigned* __restrict__ range) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    float m = (p < S.n && S.selection[p] == 0) ? S.mass[p] : 0.0f;
    unsigned lo = (m > 0.0f) ? __float_as_uint(m) : 0xffffffffu, hi = (m > 0.0f) ? __float_as_uint(m) : 0u;
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, off, 64));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(range, lo); atomicMax(range + 1, hi); }
}

__global__ void mass_kernel(MpmPtrs S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S.n) S.mass[i] = S.density[i] * S.vol[i];
}
// compute_mu_lam_from_E_nu / compute_bulk, mpm_utils.py:282-293
__global__ void mu_lam_kernel(MpmPtrs S, int with_bulk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.n) return;
    const float E = S.E[i], nu = S.nu[i];
    const float mu = E / (2.0f * (1.0f + nu));
    const float lam = E * nu / ((1.0f + nu) * (1.0f - 2.0f * nu));
    S.mu[i] = mu;
    S.lam[i] = lam;
    if (with_bulk) S.bulk[i] = lam + 2.f / 3.f * mu;
}
// apply_additional_params, mpm_utils.py:591-610
__global__ void additional_params_kernel(MpmPtrs S, float3 pt, float3 sz, float E, float nu, float density, int material) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.n) return;
    const float px = S.x[i], py = S.x[S.n + i], pz = S.x[2 * S.n + i];
    if (px > pt.x - sz.x && px < pt.x + sz.x && py > pt.y - sz.y && py < pt.y + sz.y && pz > pt.z - sz.z && pz < pt.z + sz.z) {
        S.E[i] = E; S.nu[i] = nu; S.density[i] = density; S.material[i] = material;
    }
}
// The same for a list of boxes applied in order: every particle keeps the LAST box that contains it (what N sequential
// launches of the kernel above leave behind).  Boxes are staged through LDS 256 at a time; the containment test is the
// reference's float32 expression, box by box.
__global__ __launch_bounds__(256) void additional_params_batch_kernel(MpmPtrs S, long n_boxes, const float* __restrict__ boxes,
                                                                       const float* __restrict__ params, const int* __restrict__ material) {
    __shared__ float sb[256][6];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < S.n;
    const float px = valid ? S.x[i] : 0.0f, py = valid ? S.x[S.n + i] : 0.0f, pz = valid ? S.x[2 * S.n + i] : 0.0f;
    long last = -1;
    for (long b0 = 0; b0 < n_boxes; b0 += 256) {
        const long mine = b0 + threadIdx.x;
        __syncthreads();
        if (mine < n_boxes)
            for (int k = 0; k < 6; ++k) sb[threadIdx.x][k] = boxes[mine * 6 + k];
        __syncthreads();
        const int cnt = (int)min(256L, n_boxes - b0);
        if (valid)
            for (int j = 0; j < cnt; ++j) {
                const float* q = sb[j];
                if (px > q[0] - q[3] && px < q[0] + q[3] && py > q[1] - q[4] && py < q[1] + q[4] && pz > q[2] - q[5] && pz < q[2] + q[5]) last = b0 + j;
            }
    }
    if (valid && last >= 0) {
        S.E[i] = params[last * 3]; S.nu[i] = params[last * 3 + 1]; S.density[i] = params[last * 3 + 2]; S.material[i] = material[last];
    }
}
// selection kernels, mpm_utils.py:613-663
__global__ void select_kernel(MpmPtrs S, PModDev m, float3 size, float half_height, float radius, int* mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.n) return;
    const float o[3] = {S.x[i] - m.point[0], S.x[S.n + i] - m.point[1], S.x[2 * S.n + i] - m.point[2]};
    int sel;
    if (m.type == PIXIE_PM_ROTATION) {
        const float dn = o[0] * m.normal[0] + o[1] * m.normal[1] + o[2] * m.normal[2];
        const float hx = o[0] - dn * m.normal[0], hy = o[1] - dn * m.normal[1], hz = o[2] - dn * m.normal[2];
        sel = (fabsf(dn) < half_height && sqrtf(hx * hx + hy * hy + hz * hz) < radius) ? 1 : 0;
    } else {
        sel = (fabsf(o[0]) < size.x && fabsf(o[1]) < size.y && fabsf(o[2]) < size.z) ? 1 : 0;
    }
    mask[S.perm[i]] = sel;  // masks are kept in the caller's order
}
// compute_cov_from_F, mpm_utils.py:529-553
__global__ void cov_kernel(MpmPtrs S, const float* __restrict__ init_cov, float* __restrict__ cov, const int* perm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.n) return;
    const int s = perm ? perm[i] : i;
    Mat3 F, M;
    for (int c = 0; c < 9; ++c) F.m[c] = S.Ft[(size_t)c * S.n + i];
    const float* c6 = init_cov + (size_t)s * 6;
    M.m[0] = c6[0]; M.m[1] = c6[1]; M.m[2] = c6[2];
    M.m[3] = c6[1]; M.m[4] = c6[3]; M.m[5] = c6[4];
    M.m[6] = c6[2]; M.m[7] = c6[4]; M.m[8] = c6[5];
    const Mat3 T = mat_mul_bt(mat_mul(F, M), F);
    float* o = cov + (size_t)s * 6;
    o[0] = T.m[0]; o[1] = T.m[1]; o[2] = T.m[2]; o[3] = T.m[4]; o[4] = T.m[5]; o[5] = T.m[8];
}
// Per-frame export for the rasteriser (PG/gs_simulation.py:591-600): positions and covariances of the first n_out
// particles (caller order) in the original scene frame,
//   pos_render = apply_inverse_rotations(undotransform2origin(undoshift2center111(x, z_shift), scale, mean), Rs)
//   cov_render = apply_inverse_cov_rotations(compute_cov_from_F(F_trial, init_cov) / scale^2, Rs)
// (PG/utils/transformation_utils.py:19-20,108-130; compute_cov_from_F mpm_utils.py:529-553) with the rotation chain
// folded into one matrix M on the host: pos @ M, M^T cov M.
struct FrameXform { float shift[3]; float inv_scale; float mean[3]; float inv_scale2; float M[9]; };
__global__ void frame_export_kernel(MpmPtrs S, const float* __restrict__ init_cov, FrameXform X, int n_out, float* __restrict__ pos_out,
                                    float* __restrict__ cov_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.n) return;
    const int s = S.perm[i];
    if (s >= n_out) return;
    float q[3];
    for (int d = 0; d < 3; ++d) q[d] = X.mean[d] + (S.x[(size_t)d * S.n + i] - X.shift[d]) * X.inv_scale;
    for (int d = 0; d < 3; ++d) pos_out[(size_t)s * 3 + d] = q[0] * X.M[d] + q[1] * X.M[3 + d] + q[2] * X.M[6 + d];
    if (!cov_out) return;
    Mat3 F, C0, M;
    for (int c = 0; c < 9; ++c) { F.m[c] = S.Ft[(size_t)c * S.n + i]; M.m[c] = X.M[c]; }
    const float* c6 = init_cov + (size_t)s * 6;
    C0.m[0] = c6[0]; C0.m[1] = c6[1]; C0.m[2] = c6[2];
    C0.m[3] = c6[1]; C0.m[4] = c6[3]; C0.m[5] = c6[4];
    C0.m[6] = c6[2]; C0.m[7] = c6[4]; C0.m[8] = c6[5];
    Mat3 T = mat_mul_bt(mat_mul(F, C0), F);                 // F C0 F^T
    for (int c = 0; c < 9; ++c) T.m[c] *= X.inv_scale2;
    // M^T T M
    Mat3 Mt;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) Mt.m[3 * a + b] = M.m[3 * b + a];
    const Mat3 Rr = mat_mul(mat_mul(Mt, T), M);
    float* o = cov_out + (size_t)s * 6;
    o[0] = Rr.m[0]; o[1] = Rr.m[1]; o[2] = Rr.m[2]; o[3] = Rr.m[4]; o[4] = Rr.m[5]; o[5] = Rr.m[8];
}
// compute_R_from_F, mpm_utils.py:556-580 (stores R^T)
__global__ void rot_kernel(MpmPtrs S, float* __restrict__ Rout, const int* perm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.n) return;
    const int s = perm ? perm[i] : i;
    Mat3 F, U, V;
    float sg[3];
    for (int c = 0; c < 9; ++c) F.m[c] = S.Ft[(size_t)c * S.n + i];
    svd3(F, U, sg, V);
    const Mat3 R = mat_mul_bt(U, V);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Rout[(size_t)s * 9 + 3 * a + b] = R.m[3 * b + a];
}
// export_particle_stress_to_torch (mpm_solver_warp.py:691-692): the Kirchhoff stress of the stored,
// return-mapped F -- what the last compute_stress_from_F_trial left in particle_stress.
__global__ void stress_export_kernel(MpmPtrs S, float* __restrict__ out, const int* perm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.n) return;
    const int s = perm ? perm[i] : i;
    Mat3 F;
    for (int c = 0; c < 9; ++c) F.m[c] = S.F[(size_t)c * S.n + i];
    const Mat3 tau = kirchhoff_stress(S.material[i], F, S.mu[i], S.lam[i], S.bulk[i]);
    for (int c = 0; c < 9; ++c) out[(size_t)s * 9 + c] = tau.m[c];
}
__global__ void grid_export_kernel(const float4* __restrict__ g, float* __restrict__ out, long total, int what) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float4 q = g[i];
    if (what == 0) out[i] = q.w;
    else { out[3 * i] = q.x; out[3 * i + 1] = q.y; out[3 * i + 2] = q.z; }
}

#ifdef PIXIE_DIAG
int mpm_trace_read(unsigned long long* host, int n_words) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mpm_trace), (size_t)n_words * sizeof(unsigned long long));
}
#endif
}  // namespace pixie

// ====================================================================== host side
using namespace pixie;

struct pixie_mpm {
    MpmPtrs S{};
    double grid_lim = 1.0;
    double time = 0.0;
    // host mirror of MPMModelStruct scalars (mpm_solver_warp.py:74-92)
    float g[3] = {0, 0, 0};
    float rpic = 0.0f, damping = 1.1f;
    MaterialScalars ms{};
    std::vector<pixie_bc_desc> bcs;          // host copies; cuboid points advance like `modify` (:899-905)
    std::vector<BCDev> bcs_dev;              // float versions (what the kernels see)
    std::vector<PModDev> pmods;
    std::vector<int*> masks;
    float* init_cov = nullptr;               // [n][6], caller order
    // particle word array [R_COUNT][n], two copies ping-ponged by the re-binning permutation
    unsigned* words[2] = {nullptr, nullptr};
    int cur = 0;
    // re-binning scratch
    int nblocks = 0;
    int *keys = nullptr, *rank = nullptr, *counts = nullptr, *offsets = nullptr, *order = nullptr, *order2 = nullptr, *d_n_items = nullptr;
    int4* items = nullptr;
    int* h_n_items = nullptr;                // pinned
    int n_items = 0;
    bool needs_sort = true;                  // positions changed behind the binning's back (or never binned)
    int resort_interval = 4, steps_since_sort = 0;  // starts cautious, doubles while the measured drift allows
    bool xref_valid = false;
    int trace = 0;
    bool resort_auto = true;                 // adapt resort_interval to the observed drift (off once the caller sets it)
    unsigned long long slow_at_rebin = 0;
    unsigned long long lost_seen = 0;        // oob[0] + oob[2] as read back at the last re-binning
    int2* blk_items = nullptr;               // per block: (first work item, item count)
    int occupancy = 5;                       // register-allocation target of the fused kernel (waves per SIMD): 5 or 6
    int item_cap = kWG;                      // particles per work item of the current binning: 128 or 256
    int item_cap_user = 0;                   // set_scalar "item_cap": 0 = automatic
    bool auto_half_items = false;            // item_cap "auto": 128-thread work items from the next re-binning on (see rebin())
    int comp_x = 0;                          // set_scalar "compensated_x"
    int xcd_order = 1;                       // set_scalar "xcd_order" (on: 63.5 -> 61.9 us per substep at 1 M, 18.0 -> 17.5 at 100 k, same bits; profiles/r5g_xcd_order.txt)
    bool pmods_were_active = false;
    float4* part = nullptr;                  // staged tiles of the last P2G, [max_items][kTN]
    unsigned long long* tile_mask = nullptr; // [max_items][8] occupancy bits of the staged tiles
    int grid_rb = 0;                         // grid kernel: tile loads in flight per candidate block (0 = by scene size; 1, 2, 4)
    ScanTriple* scan_chunks = nullptr;       // [ceil(nblocks / 1024)] re-binning scan scratch
    int sparse = -1;                         // sparse tile publishing: -1 auto (on when the work list exceeds two items per CU), 0, 1
    bool pending_p2g = false;                // staged tiles not yet consumed by the grid kernel
    int* blk_flags = nullptr;
    int* active_list = nullptr;              // blocks with particles in their 27-neighbourhood (built at re-binning)
    int2* nbr_table = nullptr;               // 28 int2 per active block (see MpmPtrs)
    int scatter_bits = 32;                   // mode in force: 32 = packed pairs of 32-bit sums, 2 LDS atomics per node; 64 = exact 64-bit fixed point, 4 per node
    int scatter_bits_user = 0;               // 0 (default): chosen by the particle-mass contrast at every re-binning (below); 32 / 64: forced
    bool mass_range_dirty = true;            // masses changed since the contrast was last measured
    unsigned* d_mass_range = nullptr;        // [0] min, [1] max of the positive particle masses, as float bits
    float mass_contrast = 1.0f;              // max / min as of the last measurement
    int wide = -1;                           // -1 auto: the latency-optimised variant when the scene cannot fill the chip; 0/1 forced
    int n_cus = 256;
    int n_active = 0;
    bool gout_sparse = false;                // inactive blocks of gout are stale (refreshed on export)
    StepParams last_grid_sp{};
    std::vector<BCDev> last_grid_bcs;
    long n_sorts = 0;
    std::vector<void*> allocs;
    std::vector<void*> grid_allocs;          // everything sized by n_grid (re-made by pixie_mpm_regrid)
    bool dirty_grid = false;                 // gin holds an un-consumed P2G (phase API)
    // profiling
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_particle, ev_grid;
};

namespace {

template <typename T>
int dev_alloc(pixie_mpm* h, T** ptr, size_t count, bool grid_sized = false) {
    void* p = nullptr;
    PX_CHECK_HIP(hipMalloc(&p, count * sizeof(T)));
    (grid_sized ? h->grid_allocs : h->allocs).push_back(p);
    PX_CHECK_HIP(hipMemset(p, 0, count * sizeof(T)));
    *ptr = static_cast<T*>(p);
    return 0;
}

// The grid kernel runs one wave per active block.  Its RB = 4 instantiation fits two waves per SIMD: with more active blocks
// than that (1 M particles in 120^3: ~2600 on 2048 slots) the launch takes a second round of waves and the leaner RB = 1
// instantiation + sparse tiles win (14.9 -> 12.5 us); below it (100 k in 50^3: ~520) a launch is one wave's latency and the
// fewer dependent loads the better (whole tiles, RB = 4: 18.7 vs 20.9 us per substep).
bool grid_kernel_crowded(const pixie_mpm* h) { return h->n_active > 8 * h->n_cus; }

// point the row pointers of h->S at the current copy of the word array
void bind_rows(pixie_mpm* h) {
    MpmPtrs& S = h->S;
    const size_t n = (size_t)S.n;
    float* f = reinterpret_cast<float*>(h->words[h->cur]);
    int* i = reinterpret_cast<int*>(h->words[h->cur]);
    S.x = f + R_X * n; S.v = f + R_V * n; S.F = f + R_F * n; S.Ft = f + R_FT * n; S.C = f + R_C * n;
    S.vol = f + R_VOL * n; S.mass = f + R_MASS * n; S.density = f + R_DENSITY * n; S.E = f + R_E * n; S.nu = f + R_NU * n;
    S.mu = f + R_MU * n; S.lam = f + R_LAM * n; S.bulk = f + R_BULK * n; S.ys = f + R_YS * n;
    S.material = i + R_MATERIAL * n; S.selection = i + R_SELECTION * n; S.perm = i + R_PERM * n; S.xref = f + R_XREF * n; S.xlo = f + R_XLO * n;
    S.items = h->items; S.part = h->part; S.tile_mask = h->tile_mask; S.blk_items = h->blk_items; S.blk_flags = h->blk_flags; S.active_list = h->active_list; S.nbr_table = h->nbr_table;
}

// counts -> offsets, work list, per-block item ranges (three scan launches), active blocks + their neighbour rows
void build_work_list(pixie_mpm* h, hipStream_t st) {
    const int nchunks = cdiv(h->nblocks, kScanChunk);
    hipLaunchKernelGGL(bin_scan_partial_kernel, dim3(nchunks), dim3(kScanChunk), 0, st, h->counts, h->scan_chunks, h->nblocks, h->item_cap);
    hipLaunchKernelGGL(bin_scan_chunks_kernel, dim3(1), dim3(1024), 0, st, h->scan_chunks, nchunks, h->d_n_items);
    hipLaunchKernelGGL(bin_scan_write_kernel, dim3(nchunks), dim3(kScanChunk), 0, st, h->counts, h->scan_chunks, h->offsets, h->items, h->blk_items,
                       h->nblocks, h->item_cap);
    (void)hipMemsetAsync(h->d_n_items + 1, 0, sizeof(int), st);
    hipLaunchKernelGGL(bin_mark_active_kernel, dim3(cdiv(h->nblocks, 256)), dim3(256), 0, st, h->counts, h->blk_flags, h->active_list,
                       h->d_n_items + 1, h->S.nbk, h->blk_items, h->nbr_table);  // (rewrites every flag: no slow-path writes are pending here)
}

// Re-bin the particles by grid block (counting sort) and rebuild the work list.  Everything runs on the device;
// the host only waits for the 4-byte item count so that the block kernel gets an exact grid.
int rebin(pixie_mpm* h, hipStream_t st, bool g2p_follows = false) {
    MpmPtrs& S = h->S;
    const int n = S.n;
    // Work-item capacity: 256 particles (one per thread).  128-thread items were measured too (set_scalar "item_cap"):
    // 100 k particles 21.0 -> 23.8 us per launch (916 items instead of 526: the per-item tile staging / barriers /
    // publish dominate), 1 M particles 100 -> 95 us but the grid kernel pays for twice the tiles (14.8 -> 20.7 us).
    // Automatic: 128-thread items in scenes so sparse that few blocks hold more than 128 particles -- there a 256-thread
    // workgroup runs two waves without a particle (the reference's sand configuration at 1 M: 120 -> 108 us per substep) -- and
    // 256 everywhere else (1 M in 120^3: 62.5 vs 70.9 us; 100 k in 50^3: 17.5 vs 21.8; profiles/r5e_item_cap_sparse_scenes.txt).
    // Decided from the block histogram of THIS re-binning (both item counts come back with it; if the choice flips, only the work
    // list is built again -- three small launches): see below.
    h->item_cap = h->item_cap_user > 0 ? h->item_cap_user : (h->auto_half_items ? kWG / 2 : kWG);
    PX_CHECK_HIP(hipMemsetAsync(h->counts, 0, (size_t)h->nblocks * sizeof(int), st));
    PX_CHECK_HIP(hipMemsetAsync(h->d_n_items + 2, 0, sizeof(int), st));
    hipLaunchKernelGGL(bin_count_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, S, h->keys, h->rank, h->counts,
                       reinterpret_cast<unsigned*>(h->d_n_items + 2));
    build_work_list(h, st);
    hipLaunchKernelGGL(bin_order_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, h->keys, h->rank, h->offsets, h->order, n);
    // keys/rank are free again: reuse them as the local kernel's scratch
    hipLaunchKernelGGL(bin_local_order_kernel, dim3((unsigned)h->nblocks), dim3(256), 0, st, S, h->counts, h->offsets, h->order, h->keys,
                       h->rank, h->order2);
    const int rows_per_y = 9;
    hipLaunchKernelGGL(bin_permute_kernel, dim3(cdiv(n, 256), cdiv(R_COUNT, rows_per_y)), dim3(256), 0, st,
                       h->words[h->cur], h->words[h->cur ^ 1], h->order2, n, rows_per_y, g2p_follows ? 1 : 0);
    PX_CHECK_HIP(hipGetLastError());
    const bool measure_mass = h->mass_range_dirty;
    if (measure_mass) {
        PX_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)h->d_mass_range, (int)0xffffffffu, 1, st));
        PX_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)(h->d_mass_range + 1), 0, 1, st));
        hipLaunchKernelGGL(mass_range_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, S, h->d_mass_range);
        PX_CHECK_HIP(hipMemcpyAsync(h->h_n_items + 6, h->d_mass_range, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    }
    PX_CHECK_HIP(hipMemcpyAsync(h->h_n_items, h->d_n_items, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    PX_CHECK_HIP(hipMemcpyAsync(h->h_n_items + 4, h->S.oob + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    PX_CHECK_HIP(hipMemcpyAsync(h->h_n_items + 8, h->S.oob, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    PX_CHECK_HIP(hipMemcpyAsync(h->h_n_items + 10, h->S.oob + 2, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    PX_CHECK_HIP(hipStreamSynchronize(st));
    {   // particles lost so far (left the grid / left every active block), as of this re-binning: free to read, no extra sync
        unsigned long long a, b;
        memcpy(&a, h->h_n_items + 8, sizeof a); memcpy(&b, h->h_n_items + 10, sizeof b);
        h->lost_seen = a + b;
    }
    h->n_items = h->h_n_items[0];
    h->n_active = h->h_n_items[1];
    if (h->item_cap == kWG || h->item_cap == kWG / 2) {
        // Half-size work items iff they would be nearly as few as full-size ones: enter at <= 15 % more, leave above 25 % (a scene
        // near the threshold does not flip at every re-binning; the per-item fixed-point scale depends on the item size).
        // (Break-even measured near +20 %: 1 M jelly in 200^3, +22 % items, 89.6 vs 90.6 us; the sand configuration, +2 ... 8 % over its
        // run, 120 -> 108 us; dense scenes, +64 ... 74 % items, lose 13 %.  profiles/r5e_item_cap_sparse_scenes.txt)
        const long other = h->h_n_items[3];
        const long n256 = (h->item_cap == kWG) ? h->n_items : other, n128 = (h->item_cap == kWG) ? other : h->n_items;
        if (!h->auto_half_items) h->auto_half_items = n128 * 100 <= n256 * 115;
        else h->auto_half_items = n128 * 100 <= n256 * 125;
        const int want = h->auto_half_items ? kWG / 2 : kWG;
        if (h->item_cap_user == 0 && want != h->item_cap) {   // in force from THIS binning: rebuild the work list for the other capacity
            h->item_cap = want;
            build_work_list(h, st);
            PX_CHECK_HIP(hipMemcpyAsync(h->h_n_items, h->d_n_items, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
            PX_CHECK_HIP(hipStreamSynchronize(st));
            h->n_items = h->h_n_items[0];
            h->n_active = h->h_n_items[1];
        }
    }
    if (measure_mass) {
        float lo, hi;
        memcpy(&lo, h->h_n_items + 6, sizeof lo); memcpy(&hi, h->h_n_items + 7, sizeof hi);
        h->mass_contrast = (lo > 0.0f && hi >= lo && hi < 3.0e38f) ? hi / lo : 1.0f;
        h->mass_range_dirty = false;
    }
    // (like sparse_tiles: decided at a re-binning only -- the mode changes how a tile is summed in LDS, not what is published)
    h->scatter_bits = h->scatter_bits_user ? h->scatter_bits_user : (h->mass_contrast > kPackMaxContrast ? 64 : 32);
    // (decided here and only here: never between a P2G and the grid kernel that consumes its tiles)
    S.sparse_tiles = h->sparse >= 0 ? h->sparse : (grid_kernel_crowded(h) ? 1 : 0);
    // Cadence: the LDS tile tolerates one cell of drift, and the measured drift of the interval just finished
    // predicts the next one -- aim at 0.4 cells, never more than four times the last interval (a re-binning of 1 M
    // particles costs ~0.4 ms = 4 substeps: with doubling, the ramp 4, 8, ..., 256 of a quiet scene spent six of them in
    // the first 252 substeps), and halve when > 0.1 % of the particles were on the slow path.  (The host enqueues substeps
    // far ahead of the device, so this is the only feedback.)  The 0.4 is not slack to be spent: aiming at 0.55 / 0.7 cells buys the
    // reference's sand configuration 1 / 2 % (9 -> 6 re-binnings in 400 substeps) and costs a 100 k scene in motion 34 % / 183 % -- an
    // accelerating scene overshoots the prediction and lands on the slow path (profiles/r6d_drift_target.txt).
    if (h->resort_auto) {
        unsigned long long slow_total;
        memcpy(&slow_total, h->h_n_items + 4, sizeof slow_total);
        const unsigned long long since = slow_total - h->slow_at_rebin;
        h->slow_at_rebin = slow_total;
        float d2;
        memcpy(&d2, h->h_n_items + 2, sizeof d2);
        const double drift_cells = sqrt((double)d2) * (double)S.inv_dx;
        if (h->n_sorts > 0 && h->xref_valid) {
            int k = h->resort_interval;
            if (drift_cells > 0.0) k = (int)std::min<double>(4.0 * k, std::max<double>(0.25 * k, k * 0.4 / drift_cells));
            else k = 4 * k;
            if (since > (unsigned long long)n / 1000) k = std::min(k, h->resort_interval / 2);
            // (ceiling 1024: a re-binning of 1 M particles is ~6 substeps' worth of time, so at 256 it still cost 2.6 % of a
            // quiet scene's run; a scene that wakes up inside a long interval pays through the slow path until the next one)
            h->resort_interval = std::max(2, std::min(k, 1024));
        }
        h->xref_valid = true;
    }
    h->cur ^= 1;
    bind_rows(h);
    h->needs_sort = false;
    h->steps_since_sort = 0;
    ++h->n_sorts;
    return 0;
}

StepParams make_params(const pixie_mpm* h, double dt, double time) {
    StepParams sp{};
    sp.dt = (float)dt;
    sp.time = (float)time;
    for (int d = 0; d < 3; ++d) sp.g[d] = h->g[d];
    sp.damping = h->damping;
    sp.do_damping = (h->damping < 1.0f) ? 1 : 0;  // gate mpm_solver_warp.py:595
    sp.rpic = h->rpic;
    sp.trace = h->trace;
    sp.comp_x = h->comp_x;
    sp.xcd_order = h->xcd_order;
    sp.ms = h->ms;
    return sp;
}

BCDev to_dev(const pixie_bc_desc& b) {
    BCDev d{};
    d.type = b.type; d.surface_type = b.surface_type; d.reset = b.reset;
    for (int k = 0; k < 3; ++k) {
        d.point[k] = (float)b.point[k]; d.size[k] = (float)b.size[k];
        d.velocity[k] = (float)b.velocity[k]; d.normal[k] = (float)b.normal[k];
    }
    d.start = (float)b.start_time; d.end = (float)b.end_time; d.friction = (float)b.friction;
    return d;
}

struct FieldInfo { void* ptr; int k; bool is_int; bool soa; };

bool find_field(pixie_mpm* h, const std::string& name, FieldInfo* fi) {
    MpmPtrs& S = h->S;
    struct Row { const char* nm; void* p; int k; bool is_int; };
    const Row rows[] = {
        {"x", S.x, 3, false}, {"v", S.v, 3, false}, {"F", S.F, 9, false}, {"F_trial", S.Ft, 9, false},
        {"C", S.C, 9, false}, {"vol", S.vol, 1, false}, {"mass", S.mass, 1, false}, {"density", S.density, 1, false},
        {"E", S.E, 1, false}, {"nu", S.nu, 1, false}, {"mu", S.mu, 1, false}, {"lam", S.lam, 1, false},
        {"bulk", S.bulk, 1, false}, {"yield_stress", S.ys, 1, false},
        {"material", S.material, 1, true}, {"selection", S.selection, 1, true},
    };
    for (const Row& r : rows)
        if (name == r.nm) { *fi = FieldInfo{r.p, r.k, r.is_int, true}; return true; }
    return false;
}

// The BCs of one launch, as the kernels take them
BCSet make_bcset(const pixie_mpm* h, size_t first) {
    BCSet set{};
    set.n = (int)std::min<size_t>(kMaxBCPerLaunch, h->bcs_dev.size() - std::min(first, h->bcs_dev.size()));
    for (int k = 0; k < set.n; ++k) set.bc[k] = h->bcs_dev[first + k];
    return set;
}

// host `modify` of moving cuboids (mpm_solver_warp.py:899-905) after the grid update of the substep at h->time:
// python-float maths, stored as f32
void advance_bcs(pixie_mpm* h, double dt) {
    for (size_t k = 0; k < h->bcs.size(); ++k) {
        pixie_bc_desc& b = h->bcs[k];
        if (b.type != PIXIE_BC_CUBOID) continue;
        const double t0 = (double)(float)b.start_time, t1 = (double)(float)b.end_time;
        if (h->time >= t0 && h->time < t1) {
            for (int d = 0; d < 3; ++d) {
                const float np = (float)((double)h->bcs_dev[k].point[d] + dt * (double)h->bcs_dev[k].velocity[d]);
                h->bcs_dev[k].point[d] = np;
                b.point[d] = np;
            }
        }
    }
}

template <bool G, bool P, int OCC, int FL>
void launch_block(const pixie_mpm* h, dim3 grid, hipStream_t st, const StepParams& sp, const PModSet& pms) {
    hipLaunchKernelGGL((mpm_block_kernel<G, P, OCC, FL>), grid, dim3(h->item_cap), 0, st, h->S, sp, pms);
}
// exact (64-bit) or packed (32-bit pairs) scatter
template <bool G, bool P, int OCC, int BASE>
void launch_block_p(const pixie_mpm* h, bool pack, dim3 grid, hipStream_t st, const StepParams& sp, const PModSet& pms) {
    if (pack) launch_block<G, P, OCC, BASE | F_PACK32>(h, grid, st, sp, pms);
    else launch_block<G, P, OCC, BASE>(h, grid, st, sp, pms);
}

// the fused G2P + P2G launch in the variant the scene calls for
void launch_fused_block(const pixie_mpm* h, hipStream_t st, const StepParams& sp, const PModSet& pms) {
    const dim3 grid((unsigned)std::max(h->n_items, 1));
    const bool pack = h->scatter_bits == 32;
    // Latency-optimised variant (no scheduling barriers, 130 VGPRs = three waves per SIMD = three 256-thread work items per CU at
    // once): up to that many the whole work list is resident in one round and a launch lasts one work item's latency.
    const bool wide = h->wide == 1 || (h->wide < 0 && h->n_items <= 3 * h->n_cus);
#ifdef PIXIE_DIAG
    if (h->trace & ~4) { launch_block_p<true, true, 5, F_TRACE>(h, pack, grid, st, sp, pms); return; }
#endif
    if (wide) launch_block_p<true, true, 2, F_WIDE>(h, pack, grid, st, sp, pms);
    else if (h->occupancy >= 6 && !pack) launch_block<true, true, 6, 0>(h, grid, st, sp, pms);   // six waves per SIMD: the exact scatter only (the one measured and tested)
    else launch_block_p<true, true, 5, 0>(h, pack, grid, st, sp, pms);
}

// particle modifiers whose window contains `time` (float compare, as the kernels do), impulses first (mpm_solver_warp.py:529-547)
std::vector<PModDev> active_pmods(const pixie_mpm* h, float time) {
    std::vector<PModDev> ordered;
    for (int pass = 0; pass < 2; ++pass)
        for (size_t k = 0; k < h->pmods.size(); ++k) {
            const PModDev& pm = h->pmods[k];
            if ((pm.type == PIXIE_PM_IMPULSE) != (pass == 0)) continue;
            if (!(time >= pm.start && time < pm.end)) continue;
            ordered.push_back(pm);
        }
    return ordered;
}

int launch_particle(pixie_mpm* h, bool g2p, bool p2g, const StepParams& sp, hipStream_t st) {
    // (never while staged tiles are waiting for the grid kernel: the work list they are indexed by must not change)
    if (!h->pending_p2g && (h->needs_sort || (h->resort_interval > 0 && h->steps_since_sort >= h->resort_interval)))
        if (rebin(h, st, g2p)) return 1;
    if (g2p) ++h->steps_since_sort;
    const int blocks = cdiv(h->S.n, 256);
    const dim3 grid((unsigned)std::max(h->n_items, 1));
    PModSet pms{};
    // modifiers whose time window cannot contain this substep are dropped on the host
    std::vector<PModDev> ordered = active_pmods(h, sp.time);
    if (h->resort_auto && !ordered.empty() && !h->pmods_were_active)   // a modifier switches on: velocities may jump
        h->resort_interval = std::min(h->resort_interval, 4);
    h->pmods_were_active = !ordered.empty();
    const bool fused_mods = ordered.size() <= (size_t)kMaxPModFused;
    if (fused_mods) {
        pms.n = (int)ordered.size();
        for (int k = 0; k < pms.n; ++k) pms.pm[k] = ordered[k];
    }
    if (h->n_items == 0) return 0;  // no particles binned (n_particles > 0 always gives >= 1 item)
    const bool pack = h->scatter_bits == 32;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->profile && p2g && g2p) {
        PX_CHECK_HIP(hipEventCreate(&e0)); PX_CHECK_HIP(hipEventCreate(&e1));
        PX_CHECK_HIP(hipEventRecord(e0, st));
    }
    if (g2p && p2g && fused_mods) {
        launch_fused_block(h, st, sp, pms);
    } else {
        if (g2p) {
            PModSet none{};
            launch_block<true, false, 5, 0>(h, grid, st, sp, none);
        }
        if (p2g) {
            if (!fused_mods) {
                for (const PModDev& m : ordered)
                    hipLaunchKernelGGL(pmod_kernel, dim3(blocks), dim3(256), 0, st, h->S, sp, m);
            }
            launch_block_p<false, true, 5, 0>(h, pack, grid, st, sp, pms);
        }
    }
    if (e0) {
        PX_CHECK_HIP(hipEventRecord(e1, st));
        h->ev_particle.emplace_back(e0, e1);
    }
    PX_CHECK_HIP(hipGetLastError());
    if (p2g) h->pending_p2g = true;
    return 0;
}

void launch_grid_blocks(const pixie_mpm* h, hipStream_t st, const StepParams& sp, const BCSet& set, int mode, int n_wg) {
    const int rb = h->grid_rb > 0 ? h->grid_rb : (grid_kernel_crowded(h) ? 1 : 4);
    const dim3 g((unsigned)n_wg), b(64);
    if (rb == 1) hipLaunchKernelGGL(mpm_grid_block_kernel<1>, g, b, 0, st, h->S, sp, set, mode);
    else if (rb == 2) hipLaunchKernelGGL(mpm_grid_block_kernel<2>, g, b, 0, st, h->S, sp, set, mode);
    else hipLaunchKernelGGL(mpm_grid_block_kernel<4>, g, b, 0, st, h->S, sp, set, mode);
}

int launch_grid(pixie_mpm* h, const StepParams& sp, double dt, hipStream_t st) {
    const long total = (long)h->S.ng * h->S.ng * h->S.ng;
    const int blocks = cdiv(total, 256);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->profile) {
        PX_CHECK_HIP(hipEventCreate(&e0)); PX_CHECK_HIP(hipEventCreate(&e1));
        PX_CHECK_HIP(hipEventRecord(e0, st));
    }
    const size_t nbc = h->bcs_dev.size();
    size_t done = 0;
    int normalise = 1;
    do {
        const BCSet set = make_bcset(h, done);
        if (normalise && h->pending_p2g && nbc <= (size_t)kMaxBCPerLaunch) {
            // staged tiles of the last P2G + slow-path atomics in gin; blocks with nothing nearby are skipped
            launch_grid_blocks(h, st, sp, set, 0, std::max(h->n_active, 1));
            h->gout_sparse = true;
            h->last_grid_sp = sp;
            h->last_grid_bcs.assign(h->bcs_dev.begin(), h->bcs_dev.end());
        } else if (normalise && h->pending_p2g) {  // more BCs than one launch carries: dense follow-up passes need every block
            launch_grid_blocks(h, st, sp, set, 0, std::max(h->n_active, 1));
            launch_grid_blocks(h, st, sp, set, 1, h->nblocks);
            h->gout_sparse = false;
        } else {                          // nothing staged (or a further pass of BCs over gout)
            hipLaunchKernelGGL(mpm_grid_kernel, dim3(blocks), dim3(256), 0, st, h->S, sp, set, normalise);
            if (normalise) h->gout_sparse = false;
        }
        done += set.n;
        normalise = 0;
    } while (done < nbc);
    h->pending_p2g = false;
    if (e0) {
        PX_CHECK_HIP(hipEventRecord(e1, st));
        h->ev_grid.emplace_back(e0, e1);
    }
    PX_CHECK_HIP(hipGetLastError());
    advance_bcs(h, dt);
    return 0;
}

// Everything whose size depends on n_grid: the two grid arrays, the block tables and the work list / staged tiles.
// Used by pixie_mpm_create and by pixie_mpm_regrid (set_parameters_dict changing n_grid / grid_lim after the particles
// were loaded, mpm_solver_warp.py:315-342: the reference re-allocates the grids and recomputes dx, nothing else).
// The new buffers are allocated first; the old ones are released only when every allocation succeeded, so a failed
// regrid leaves the handle as it was.
int alloc_grid(pixie_mpm* h, int n_grid, double grid_lim) {
    MpmPtrs& S = h->S;
    const int nbk = (n_grid + kBS - 1) / kBS;
    const int nblocks = nbk * nbk * nbk;
    const size_t n = (size_t)S.n, G = (size_t)n_grid * n_grid * n_grid;
    const size_t max_items = (n + 63) / 64 + std::min<size_t>((size_t)nblocks, n);   // for the smallest capacity (64)
    // gather_node (grid kernel) forms tile offsets item * kTN in 32-bit arithmetic: 8.4 M work items, i.e. > 100 M particles
    // at the smallest capacity, would wrap silently -- refused here instead
    PX_REQUIRE(max_items * (size_t)kTN < ((size_t)1 << 32), "pixie_mpm: %zu work items exceed the 32-bit tile offsets of the grid kernel (2^23 items)", max_items);
    std::vector<void*> old;
    old.swap(h->grid_allocs);
    float4 *gin = nullptr, *gout = nullptr, *part = nullptr;
    unsigned long long* tile_mask = nullptr;
    int *counts = nullptr, *offsets = nullptr, *active_list = nullptr, *blk_flags = nullptr;
    int4* items = nullptr;
    int2 *nbr_table = nullptr, *blk_items = nullptr;
    int rc = 0;
    rc |= dev_alloc(h, &gin, G, true); rc |= dev_alloc(h, &gout, G, true);
    rc |= dev_alloc(h, &counts, (size_t)nblocks, true); rc |= dev_alloc(h, &offsets, (size_t)nblocks, true);
    rc |= dev_alloc(h, &items, max_items, true);
    rc |= dev_alloc(h, &active_list, (size_t)nblocks, true);
    rc |= dev_alloc(h, &nbr_table, (size_t)nblocks * 28, true);
    rc |= dev_alloc(h, &blk_items, (size_t)nblocks, true); rc |= dev_alloc(h, &blk_flags, (size_t)nblocks, true);
    ScanTriple* scan_chunks = nullptr;
    rc |= dev_alloc(h, &scan_chunks, (size_t)cdiv(nblocks, kScanChunk), true);
    rc |= dev_alloc(h, &part, (max_items + 1) * kTN, true);     // (+ the zero tile of gather_node; dev_alloc clears, nothing ever writes it)
    rc |= dev_alloc(h, &tile_mask, max_items * 8, true);
    if (rc) {   // keep the old grid
        for (void* p : h->grid_allocs) (void)hipFree(p);
        h->grid_allocs.swap(old);
        return 1;
    }
    for (void* p : old) (void)hipFree(p);
    S.ng = n_grid;
    h->grid_lim = grid_lim;
    S.dx = (float)(grid_lim / n_grid);              // mpm_solver_warp.py:62-66, :320-326
    S.inv_dx = (float)((double)n_grid / grid_lim);
    S.nbk = nbk;
    h->nblocks = nblocks;
    S.gin = gin; S.gout = gout;
    h->counts = counts; h->offsets = offsets; h->items = items; h->active_list = active_list; h->nbr_table = nbr_table;
    h->blk_items = blk_items; h->blk_flags = blk_flags; h->part = part; h->tile_mask = tile_mask;
    S.zero_off = (unsigned)(max_items * kTN);
    h->scan_chunks = scan_chunks;
    h->n_items = 0; h->n_active = 0;
    h->auto_half_items = false;
    h->needs_sort = true; h->xref_valid = false;
    h->pending_p2g = false; h->dirty_grid = false; h->gout_sparse = false;
    if (h->resort_auto) h->resort_interval = 4;
    return rc;
}

}  // namespace

extern "C" {

// mpm_solver_warp.py:84-86, 391-393: the reference evaluates `wp.sin` / `wp.sqrt` from Python scope, i.e. Warp's float32
// built-ins (float32 argument and result); the products around them are Python doubles and the struct member is float32.
static float host_alpha(double friction_angle) {
    const double sin_phi = (double)sinf((float)(friction_angle / 180.0 * 3.14159265));
    return (float)((double)sqrtf((float)(2.0 / 3.0)) * 2.0 * sin_phi / (3.0 - sin_phi));
}

int pixie_mpm_create(pixie_mpm** out, int n_particles, int n_grid, double grid_lim) {
    PX_REQUIRE(out && n_particles > 0 && n_grid >= 4 && grid_lim > 0, "pixie_mpm_create: bad arguments");
    pixie_mpm* h = new pixie_mpm();
    MpmPtrs& S = h->S;
    S.n = n_particles;
    const size_t n = (size_t)n_particles;
    int rc = 0;
    rc |= dev_alloc(h, &h->words[0], (size_t)R_COUNT * n); rc |= dev_alloc(h, &h->words[1], (size_t)R_COUNT * n);
    rc |= dev_alloc(h, &S.oob, 3);
    {
        unsigned* lut = nullptr;
        rc |= dev_alloc(h, &lut, (size_t)kWG);
        if (!rc) {
            unsigned host[kWG];
            for (int t = 0; t < kWG; ++t) {
                const int a = t, b = t + kWG;
                host[t] = (unsigned)staged_index(a >> 6, (a >> 3) & 7, a & 7) | ((unsigned)staged_index(b >> 6, (b >> 3) & 7, b & 7) << 16);
            }
            if (hipMemcpy(lut, host, sizeof host, hipMemcpyHostToDevice) != hipSuccess) rc = 1;
        }
        S.staged_lut = lut;
    }
    rc |= dev_alloc(h, &h->keys, n); rc |= dev_alloc(h, &h->rank, n); rc |= dev_alloc(h, &h->order, n); rc |= dev_alloc(h, &h->order2, n);
    rc |= dev_alloc(h, &h->d_n_items, 4);
    rc |= dev_alloc(h, &h->d_mass_range, 2);
    rc |= alloc_grid(h, n_grid, grid_lim);
    rc |= dev_alloc(h, &h->init_cov, 6 * n);
    if (rc) { pixie_mpm_destroy(h); return 1; }
    if (hipHostMalloc((void**)&h->h_n_items, 12 * sizeof(int)) != hipSuccess) { pixie_mpm_destroy(h); return set_error("hipHostMalloc failed"); }
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) h->n_cus = cus;
    }
    bind_rows(h);
    hipLaunchKernelGGL(iota_kernel, dim3(cdiv(n, 256)), dim3(256), 0, 0, S.perm, n_particles);
    hipLaunchKernelGGL(identity_F_kernel, dim3(cdiv(n, 256)), dim3(256), 0, 0, S.Ft, n_particles);  // :272-277
    PX_CHECK_HIP(hipDeviceSynchronize());
    // defaults of initialize(), mpm_solver_warp.py:74-92
    h->ms.plastic_viscosity = 0.0f; h->ms.softening = 0.1f; h->ms.hardening = 0.0f; h->ms.xi = 0.0f;
    h->ms.alpha = host_alpha(25.0);
    h->rpic = 0.0f; h->damping = 1.1f;
    *out = h;
    return 0;
}

int pixie_mpm_destroy(pixie_mpm* h) {
    if (!h) return 0;
    for (void* p : h->allocs) (void)hipFree(p);
    for (void* p : h->grid_allocs) (void)hipFree(p);
    for (int* m : h->masks) (void)hipFree(m);
    if (h->h_n_items) (void)hipHostFree(h->h_n_items);
    for (auto& e : h->ev_particle) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto& e : h->ev_grid) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    delete h;
    return 0;
}

int pixie_mpm_regrid(pixie_mpm* h, int n_grid, double grid_lim, void* stream) {
    PX_REQUIRE(h && n_grid >= 4 && grid_lim > 0, "pixie_mpm_regrid: bad arguments");
    PX_REQUIRE(!h->dirty_grid, "pixie_mpm_regrid: a phase-API P2G is pending");
    PX_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));   // nothing may still be reading the old grid
    if (alloc_grid(h, n_grid, grid_lim)) return 1;
    bind_rows(h);
    hipLaunchKernelGGL(unfreeze_kernel, dim3(cdiv(h->S.n, 256)), dim3(256), 0, as_stream(stream), h->S.selection, h->S.n);
    h->mass_range_dirty = true;   // re-admitted particles count towards the mass contrast again
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

int pixie_mpm_set_field(pixie_mpm* h, const char* name, const void* d_src, int64_t count, void* stream) {
    PX_REQUIRE(h && name && d_src, "pixie_mpm_set_field: null argument");
    hipStream_t st = as_stream(stream);
    const std::string nm(name);
    const int n = h->S.n;
    if (nm == "init_cov") {
        PX_REQUIRE(count == (int64_t)n * 6, "set_field(init_cov): expected %lld scalars, got %lld", (long long)n * 6, (long long)count);
        PX_CHECK_HIP(hipMemcpyAsync(h->init_cov, d_src, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToDevice, st));
        return 0;
    }
    FieldInfo fi;
    PX_REQUIRE(find_field(h, nm, &fi), "set_field: unknown field '%s'", name);
    PX_REQUIRE(count == (int64_t)n * fi.k, "set_field(%s): expected %lld scalars, got %lld", name, (long long)n * fi.k, (long long)count);
    if (nm == "mass" || nm == "selection") { h->mass_range_dirty = true; h->needs_sort = true; }
    if (nm == "material") h->needs_sort = true;     // the order inside a block goes by material class
    if (nm == "x") {   // positions replaced: binning stale, frozen particles get another chance
        h->needs_sort = true; h->xref_valid = false; h->resort_interval = h->resort_auto ? 4 : h->resort_interval;
        h->auto_half_items = false;   // a new scene decides its work-item capacity afresh, like a new handle
        hipLaunchKernelGGL(unfreeze_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, h->S.selection, n);
        h->mass_range_dirty = true;   // re-admitted particles count towards the mass contrast again
        PX_CHECK_HIP(hipMemsetAsync(h->S.xlo, 0, (size_t)3 * n * sizeof(float), st));   // new positions carry no remainder
    }
    if (fi.is_int)
        hipLaunchKernelGGL(aos_to_soa_kernel<int>, dim3(cdiv(n, 256)), dim3(256), 0, st, (const int*)d_src, (int*)fi.ptr, n, fi.k, h->S.perm);
    else
        hipLaunchKernelGGL(aos_to_soa_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, st, (const float*)d_src, (float*)fi.ptr, n, fi.k, h->S.perm);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

int pixie_mpm_get_field(pixie_mpm* h, const char* name, void* d_dst, int64_t count, void* stream) {
    PX_REQUIRE(h && name && d_dst, "pixie_mpm_get_field: null argument");
    hipStream_t st = as_stream(stream);
    const std::string nm(name);
    const int n = h->S.n;
    const long G = (long)h->S.ng * h->S.ng * h->S.ng;
    if (nm == "grid_m" || nm == "grid_v_in" || nm == "grid_v_out") {
        const int k = (nm == "grid_m") ? 1 : 3;
        PX_REQUIRE(count == (int64_t)G * k, "get_field(%s): expected %lld scalars, got %lld", name, (long long)G * k, (long long)count);
        const float4* src = (nm == "grid_v_out") ? h->S.gout : h->S.gin;
        if (nm == "grid_v_out" && h->gout_sparse) {  // bring the skipped (massless, far from any particle) blocks up to date
            BCSet set{};
            set.n = (int)h->last_grid_bcs.size();
            for (int k = 0; k < set.n; ++k) set.bc[k] = h->last_grid_bcs[k];
            launch_grid_blocks(h, st, h->last_grid_sp, set, 1, h->nblocks);
            h->gout_sparse = false;
        }
        if (nm != "grid_v_out" && h->pending_p2g)
            hipLaunchKernelGGL(grid_export_pending_kernel, dim3((unsigned)h->nblocks), dim3(64), 0, st, h->S, (float*)d_dst, nm == "grid_m" ? 0 : 1);
        else
            hipLaunchKernelGGL(grid_export_kernel, dim3(cdiv(G, 256)), dim3(256), 0, st, src, (float*)d_dst, G, nm == "grid_m" ? 0 : 1);
        PX_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (nm == "stress") {
        PX_REQUIRE(count == (int64_t)n * 9, "get_field(stress): expected %lld scalars", (long long)n * 9);
        hipLaunchKernelGGL(stress_export_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, h->S, (float*)d_dst, h->S.perm);
        PX_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (nm == "init_cov") {
        PX_REQUIRE(count == (int64_t)n * 6, "get_field(init_cov): expected %lld scalars", (long long)n * 6);
        PX_CHECK_HIP(hipMemcpyAsync(d_dst, h->init_cov, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToDevice, st));
        return 0;
    }
    if (nm == "cov") return pixie_mpm_export_cov(h, (float*)d_dst, stream);
    FieldInfo fi;
    PX_REQUIRE(find_field(h, nm, &fi), "get_field: unknown field '%s'", name);
    PX_REQUIRE(count == (int64_t)n * fi.k, "get_field(%s): expected %lld scalars, got %lld", name, (long long)n * fi.k, (long long)count);
    if (fi.is_int)
        hipLaunchKernelGGL(soa_to_aos_kernel<int>, dim3(cdiv(n, 256)), dim3(256), 0, st, (const int*)fi.ptr, (int*)d_dst, n, fi.k, h->S.perm);
    else
        hipLaunchKernelGGL(soa_to_aos_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, st, (const float*)fi.ptr, (float*)d_dst, n, fi.k, h->S.perm);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

int pixie_mpm_fill_field(pixie_mpm* h, const char* name, double value, void* stream) {
    PX_REQUIRE(h && name, "pixie_mpm_fill_field: null argument");
    FieldInfo fi;
    PX_REQUIRE(find_field(h, name, &fi), "fill_field: unknown field '%s'", name);
    const long cnt = (long)h->S.n * fi.k;
    if (fi.is_int)
        hipLaunchKernelGGL(fill_kernel<int>, dim3(cdiv(cnt, 256)), dim3(256), 0, as_stream(stream), (int*)fi.ptr, cnt, (int)value);
    else
        hipLaunchKernelGGL(fill_kernel<float>, dim3(cdiv(cnt, 256)), dim3(256), 0, as_stream(stream), (float*)fi.ptr, cnt, (float)value);
    PX_CHECK_HIP(hipGetLastError());
    if (std::string(name) == "material") h->needs_sort = true;
    return 0;
}

int pixie_mpm_set_scalar(pixie_mpm* h, const char* key, double value) {
    PX_REQUIRE(h && key, "pixie_mpm_set_scalar: null argument");
    const std::string k(key);
    if (k == "rpic_damping") h->rpic = (float)value;
    else if (k == "grid_v_damping_scale") h->damping = (float)value;
    else if (k == "hardening") h->ms.hardening = (float)value;
    else if (k == "xi") h->ms.xi = (float)value;
    else if (k == "softening") h->ms.softening = (float)value;
    else if (k == "plastic_viscosity") h->ms.plastic_viscosity = (float)value;
    else if (k == "friction_angle") h->ms.alpha = host_alpha(value);  // mpm_solver_warp.py:390-393
    else if (k == "gx") h->g[0] = (float)value;
    else if (k == "gy") h->g[1] = (float)value;
    else if (k == "gz") h->g[2] = (float)value;
    else if (k == "time") h->time = value;
#ifdef PIXIE_DIAG
    else if (k == "profile") h->profile = value != 0.0;
#else
    else if (k == "profile" || k == "trace") {   // switching them OFF is a no-op everywhere (callers written against the round-4 library do that)
        if (value != 0.0) return set_error("pixie_mpm_set_scalar(%s): diagnostic switch, only in the PIXIE_DIAG build (libpixie_hip_diag.so)", key);
    }
#endif
    else if (k == "scatter_bits") {
        PX_REQUIRE(value == 64 || value == 32 || value == 0, "scatter_bits must be 0 (auto: by mass contrast), 64 (exact) or 32 (packed pairs)");
        h->scatter_bits_user = (int)value;
        if (value != 0) h->scatter_bits = (int)value; else { h->mass_range_dirty = true; h->needs_sort = true; }
    }
    else if (k == "grid_rb") { PX_REQUIRE(value == 0 || value == 1 || value == 2 || value == 4, "grid_rb must be 0, 1, 2 or 4"); h->grid_rb = (int)value; }
    else if (k == "sparse_tiles") { PX_REQUIRE(value == -1 || value == 0 || value == 1, "sparse_tiles must be -1 (auto), 0 or 1"); h->sparse = (int)value; h->needs_sort = true; }
    else if (k == "wide") { PX_REQUIRE(value == -1 || value == 0 || value == 1, "wide must be -1 (auto), 0 or 1"); h->wide = (int)value; }
#ifdef PIXIE_DIAG
    else if (k == "trace") h->trace = (int)value;
#endif
    else if (k == "compensated_x") h->comp_x = value != 0.0 ? 1 : 0;
    else if (k == "xcd_order") h->xcd_order = value != 0.0 ? 1 : 0;
    else if (k == "occupancy") { PX_REQUIRE(value == 5 || value == 6, "occupancy must be 5 or 6 waves per SIMD"); h->occupancy = (int)value; }
    else if (k == "item_cap") { PX_REQUIRE(value == 0 || value == 64 || value == 128 || value == 192 || value == 256, "item_cap must be 0 (auto), 64, 128, 192 or 256"); h->item_cap_user = (int)value; h->needs_sort = true; }
    else if (k == "resort_interval") { h->resort_interval = (int)value; h->resort_auto = false; }   // substeps between re-binnings (0 = only when positions are replaced)
    else return set_error("set_scalar: unknown key '%s'", key);
    return 0;
}

int pixie_mpm_get_scalar(pixie_mpm* h, const char* key, double* value) {
    PX_REQUIRE(h && key && value, "pixie_mpm_get_scalar: null argument");
    const std::string k(key);
    if (k == "time") *value = h->time;
    else if (k == "dx") *value = h->S.dx;
    else if (k == "inv_dx") *value = h->S.inv_dx;
    else if (k == "alpha") *value = h->ms.alpha;
    else if (k == "rpic_damping") *value = h->rpic;
    else if (k == "grid_v_damping_scale") *value = h->damping;
    else if (k == "resort_interval") *value = h->resort_interval;
    else if (k == "n_work_items") *value = h->n_items;
    else if (k == "item_cap") *value = h->item_cap;                    // the capacity in force (as of the last re-binning when auto)
    else if (k == "scatter_bits") *value = h->scatter_bits;            // the mode in force (as of the last re-binning when auto)
    else if (k == "scatter_bits_user") *value = h->scatter_bits_user;
    else if (k == "mass_contrast") *value = h->mass_contrast;
    else if (k == "n_active_blocks") *value = h->n_active;
    else if (k == "n_rebins") *value = (double)h->n_sorts;
    else if (k == "lost_particles_seen") *value = (double)h->lost_seen;   // as of the last re-binning; does not synchronise
    else if (k == "dropped_particles") {  // slow-path particles that had left every active block; synchronises the device
        unsigned long long v = 0;
        PX_CHECK_HIP(hipMemcpy(&v, h->S.oob + 2, sizeof v, hipMemcpyDeviceToHost));
        *value = (double)v;
    }
    else if (k == "slow_path_particles") {  // synchronises the device
        unsigned long long v = 0;
        PX_CHECK_HIP(hipMemcpy(&v, h->S.oob + 1, sizeof v, hipMemcpyDeviceToHost));
        *value = (double)v;
    }
    else return set_error("get_scalar: unknown key '%s'", key);
    return 0;
}

int pixie_mpm_update_mass(pixie_mpm* h, void* stream) {
    PX_REQUIRE(h, "null handle");
    h->mass_range_dirty = true; h->needs_sort = true;
    hipLaunchKernelGGL(mass_kernel, dim3(cdiv(h->S.n, 256)), dim3(256), 0, as_stream(stream), h->S);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

int pixie_mpm_finalize_mu_lam(pixie_mpm* h, int with_bulk, void* stream) {
    PX_REQUIRE(h, "null handle");
    hipLaunchKernelGGL(mu_lam_kernel, dim3(cdiv(h->S.n, 256)), dim3(256), 0, as_stream(stream), h->S, with_bulk);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

int pixie_mpm_apply_additional_params(pixie_mpm* h, const double point[3], const double size[3], double E, double nu,
                                      double density, int material, void* stream) {
    PX_REQUIRE(h && point && size, "null argument");
    const float3 pt = make_float3((float)point[0], (float)point[1], (float)point[2]);
    const float3 sz = make_float3((float)size[0], (float)size[1], (float)size[2]);
    hipLaunchKernelGGL(additional_params_kernel, dim3(cdiv(h->S.n, 256)), dim3(256), 0, as_stream(stream), h->S, pt, sz,
                       (float)E, (float)nu, (float)density, material);
    PX_CHECK_HIP(hipGetLastError());
    h->needs_sort = true;      // materials changed: the order inside a block goes by material class
    return 0;
}

int pixie_mpm_apply_additional_params_batch(pixie_mpm* h, int64_t n_boxes, const float* d_boxes, const float* d_params,
                                            const int32_t* d_material, void* stream) {
    PX_REQUIRE(h && n_boxes >= 0 && (n_boxes == 0 || (d_boxes && d_params && d_material)), "apply_additional_params_batch: bad arguments");
    if (n_boxes == 0) return 0;
    hipLaunchKernelGGL(additional_params_batch_kernel, dim3(cdiv(h->S.n, 256)), dim3(256), 0, as_stream(stream), h->S, (long)n_boxes, d_boxes,
                       d_params, reinterpret_cast<const int*>(d_material));
    PX_CHECK_HIP(hipGetLastError());
    h->needs_sort = true;      // materials changed: the order inside a block goes by material class
    return 0;
}

int pixie_mpm_add_bc(pixie_mpm* h, const pixie_bc_desc* bc) {
    PX_REQUIRE(h && bc, "null argument");
    PX_REQUIRE(bc->type >= 0 && bc->type <= 2, "add_bc: unknown type %d", bc->type);
    h->bcs.push_back(*bc);
    h->bcs_dev.push_back(to_dev(*bc));
    return 0;
}

int pixie_mpm_add_particle_modifier(pixie_mpm* h, const pixie_pmod_desc* pm, void* stream) {
    PX_REQUIRE(h && pm, "null argument");
    PX_REQUIRE(pm->type >= 0 && pm->type <= 2, "add_particle_modifier: unknown type %d", pm->type);
    PModDev m{};
    m.type = pm->type;
    for (int d = 0; d < 3; ++d) {
        m.point[d] = (float)pm->point[d]; m.force[d] = (float)pm->force[d]; m.velocity[d] = (float)pm->velocity[d];
        m.normal[d] = (float)pm->normal[d]; m.h1[d] = (float)pm->h1[d]; m.h2[d] = (float)pm->h2[d];
    }
    m.rot_scale = (float)pm->rotation_scale; m.trans_scale = (float)pm->translation_scale;
    m.start = (float)pm->start_time; m.end = (float)pm->end_time;
    int* mask = nullptr;
    PX_CHECK_HIP(hipMalloc(&mask, (size_t)h->S.n * sizeof(int)));
    h->masks.push_back(mask);
    m.mask = mask;
    const float3 size = make_float3((float)pm->size[0], (float)pm->size[1], (float)pm->size[2]);
    hipLaunchKernelGGL(select_kernel, dim3(cdiv(h->S.n, 256)), dim3(256), 0, as_stream(stream), h->S, m, size,
                       (float)pm->half_height, (float)pm->radius, mask);
    PX_CHECK_HIP(hipGetLastError());
    h->pmods.push_back(m);
    return 0;
}

// (Round 4 captured the steady part of this loop -- 32 substeps of grid kernel + fused block kernel -- as a HIP graph, replayed
// whenever nothing the kernels compute depended on the time inside the chunk.  Bit-identical, and no faster: 18.00 vs 18.10 us
// per substep at 100 k particles, 63.8 vs 63.2 at 1 M; with several scenes sharing the GPU it was slower (two 1 M scenes: 128.8
// vs 109.8 us per pair of substeps).  The host already runs far ahead of the device, so what a substep pays at its two kernel
// boundaries is the device's own 3.6 us, graph or not.  profiles/r4h_mpm_step_graph_rejected.txt)
int pixie_mpm_step(pixie_mpm* h, double dt, int n_substeps, void* stream) {
    PX_REQUIRE(h && n_substeps >= 0, "pixie_mpm_step: bad arguments");
    if (n_substeps == 0) return 0;
    PX_REQUIRE(!h->dirty_grid, "pixie_mpm_step: a phase-API P2G is pending; finish the substep with phases 1,2 first");
    hipStream_t st = as_stream(stream);
    // substep 0: modifiers + stress + P2G at time t0
    if (launch_particle(h, false, true, make_params(h, dt, h->time), st)) return 1;
    for (int i = 0; i < n_substeps; ++i) {
        if (launch_grid(h, make_params(h, dt, h->time), dt, st)) return 1;
        h->time = h->time + dt;  // mpm_solver_warp.py:637
        const bool last = (i == n_substeps - 1);
        // G2P of substep i fused with modifiers/stress/P2G of substep i+1 (evaluated at the new time)
        if (launch_particle(h, true, !last, make_params(h, dt, h->time), st)) return 1;
    }
    return 0;
}

#ifdef PIXIE_DIAG
int pixie_mpm_phase(pixie_mpm* h, int phase, double dt, void* stream) {
    PX_REQUIRE(h, "null handle");
    hipStream_t st = as_stream(stream);
    const StepParams sp = make_params(h, dt, h->time);
    if (phase == 0) {
        PX_REQUIRE(!h->dirty_grid, "phase 0 called twice without a grid update");
        h->dirty_grid = true;
        return launch_particle(h, false, true, sp, st);
    } else if (phase == 1) {
        h->dirty_grid = false;
        return launch_grid(h, sp, dt, st);
    } else if (phase == 2) {
        return launch_particle(h, true, false, sp, st);
    }
    return set_error("pixie_mpm_phase: unknown phase %d", phase);
}
#endif

int pixie_mpm_export_cov(pixie_mpm* h, float* d_cov, void* stream) {
    PX_REQUIRE(h && d_cov, "null argument");
    hipLaunchKernelGGL(cov_kernel, dim3(cdiv(h->S.n, 256)), dim3(256), 0, as_stream(stream), h->S, h->init_cov, d_cov, h->S.perm);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

int pixie_mpm_export_frame(pixie_mpm* h, int n_out, const double shift[3], double scale, const double mean[3],
                           const double inv_rotation[9], float* d_pos, float* d_cov, void* stream) {
    PX_REQUIRE(h && shift && mean && inv_rotation && d_pos, "pixie_mpm_export_frame: null argument");
    PX_REQUIRE(n_out > 0 && n_out <= h->S.n && scale != 0.0, "pixie_mpm_export_frame: bad n_out / scale");
    FrameXform X{};
    for (int d = 0; d < 3; ++d) { X.shift[d] = (float)shift[d]; X.mean[d] = (float)mean[d]; }
    X.inv_scale = (float)(1.0 / scale); X.inv_scale2 = (float)(1.0 / (scale * scale));
    for (int c = 0; c < 9; ++c) X.M[c] = (float)inv_rotation[c];
    hipLaunchKernelGGL(frame_export_kernel, dim3(cdiv(h->S.n, 256)), dim3(256), 0, as_stream(stream), h->S, h->init_cov, X, n_out, d_pos, d_cov);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

int pixie_mpm_export_R(pixie_mpm* h, float* d_R, void* stream) {
    PX_REQUIRE(h && d_R, "null argument");
    hipLaunchKernelGGL(rot_kernel, dim3(cdiv(h->S.n, 256)), dim3(256), 0, as_stream(stream), h->S, d_R, h->S.perm);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

int pixie_mpm_out_of_bounds(pixie_mpm* h, int64_t* count, void* stream) {
    PX_REQUIRE(h && count, "null argument");
    unsigned long long v[3] = {0, 0, 0};
    PX_CHECK_HIP(hipMemcpyAsync(v, h->S.oob, sizeof v, hipMemcpyDeviceToHost, as_stream(stream)));
    PX_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    *count = (int64_t)(v[0] + v[2]);  // left the grid + left every active block (see p2g_scatter_global)
    return 0;
}

#ifdef PIXIE_DIAG
int pixie_mpm_kernel_times(pixie_mpm* h, double* particle_ms, double* grid_ms, int64_t* n_launches) {
    PX_REQUIRE(h && particle_ms && grid_ms && n_launches, "null argument");
    double tp = 0.0, tg = 0.0;
    for (auto& e : h->ev_particle) {
        PX_CHECK_HIP(hipEventSynchronize(e.second));
        float ms = 0.f;
        PX_CHECK_HIP(hipEventElapsedTime(&ms, e.first, e.second));
        tp += ms;
        (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
    }
    for (auto& e : h->ev_grid) {
        PX_CHECK_HIP(hipEventSynchronize(e.second));
        float ms = 0.f;
        PX_CHECK_HIP(hipEventElapsedTime(&ms, e.first, e.second));
        tg += ms;
        (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
    }
    *n_launches = (int64_t)h->ev_particle.size();
    *particle_ms = h->ev_particle.empty() ? 0.0 : tp / (double)h->ev_particle.size();
    *grid_ms = h->ev_grid.empty() ? 0.0 : tg / (double)h->ev_grid.size();
    h->ev_particle.clear();
    h->ev_grid.clear();
    return 0;
}
#endif

}  // extern "C"
