// pixie_amd/csrc/conv3d_f16x3.hip -- the U-Net's 3D convolutions on the gfx950 f16 matrix cores at fp32-grade
// accuracy ("f16x3": every fp32 operand is split into two fp16 halves and three MFMAs replace one).
//
// Why: the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, conv3d_mfma.hip) runs at the fp32 vector rate, 157 TFLOP/s;
// v_mfma_f32_32x32x16_f16 runs at 2.5 PFLOP/s.  With  x*s = hi + lo  (hi = fp16(x*s), lo = fp16(x*s - hi), s a
// power of two that places the tensor just below the fp16 range, so hi+lo carries 22 significant bits),
//     w.x  ~  ( w_hi.x_hi + w_hi.x_lo + w_lo.x_hi ) / (s_w s_x)
// drops only the lo.lo term (2^-22 relative) and accumulates in fp32 inside the MFMA: three f16 MFMAs per
// fp32-equivalent product = 833 TFLOP/s of algorithmic peak, 5.3x the exact-fp32 pipe, at 2-3e-7 relative error
// per product (fp32's own is 6e-8).  The U-Net parity target (<= 1e-4 rel-L2 vs the fp32 reference) is kept with
// >100x margin; tests/test_unet_hip.py measures it.
//
// Replaces the same reference ops as conv3d_mfma.hip (WG/models/module/diffusion_network.py conv_nd at
// :679,:683,:691,:762,:58,:206,:208,:872, FeatureProjector :570-583) for stride-1 layers whose channel counts are
// multiples of 16 (stride 2 for the 3^3 Downsample convs); everything else (tiny test networks) stays on the exact-fp32 kernel.
//
// Formulation: Out[co][v] = sum_{tap,ci} W[tap][ci][co] X[ci][v+tap], implicit GEMM with M = c_out (A operand),
// N = voxels (B operand, 32 x-contiguous voxels per MFMA column block), K = taps*c_in walked tap by tap in steps
// of 16 channels (one MFMA K).  Workgroup = 4 waves, each MB*32 c_out x NB*32 voxels.
//   B: per 16-channel chunk the halo'd voxel tile is staged once in LDS as fp16 hi/lo planes laid out
//      [k-group of 8 channels][voxel] x 16 B, so a wave's B fragment is one conflict-free ds_read_b128 and the
//      same tile serves all 27 taps.  Prologue (norm affine, spatial LayerNorm affine, activation, zero padding
//      AFTER the activation, nearest x2 upsampling, channel concat) is applied on the way in, as in the fp32 kernel.
//   A: weights are pre-split and pre-swizzled on the device (pack kernel below) into [tap][k-group][c_out] x 16 B
//      hi/lo planes; every wave reads its A fragments straight from L2 (two coalesced 512 B segments per load;
//      the whole 64->64 3^3 layer is 442 KB), so LDS holds activations only and two workgroups fit per CU.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <string>
#include <type_traits>

#include "../../include/pixie_hip.h"
#include "common.h"

namespace pixie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kW16HeaderU4 = 4;  // 64-byte header in front of the packed planes: [0].x = bits of 1/s_w

struct Conv16Args {
    const float* in0; const float* in1;
    int c0, cin;
    int ID, IH, IW;          // stored input dims
    int LD, LH, LW;          // logical input dims (after optional nearest x2)
    int ups;
    int stride;              // 1, or 2 (3^3 Downsample convs: the LDS tile then holds every input voxel, the B reads skip one)
    int OD, OH, OW;
    const float* pro_a; const float* pro_b; const float* gamma; const float* beta;
    int act;
    const uint4* w16;        // packed weights (header + hi planes + lo planes)
    const float* wf;         // EXACT variant: fp32 weights [tap][c_in][c_out padded] (pixie_conv_pack_weights) instead of w16
    const float* bias;
    int cout, coutp;
    const float* residual; float* out;
    const unsigned* amax0; const unsigned* amax1;   // device |x|max of in0 / in1 as float bits, or null
    float in_bound;                                  // host bound on |prologue(x)| when amax0 is null
    int TX, TY, TZ, lTX, lTY;
    int tiles_x, tiles_y, tiles_z, n_tiles;
    int HX, HY, HZ, HYX, CS;
    unsigned mHX, mHYX;
    float* stats; unsigned* out_amax;   // epilogue statistics (see the epilogue), or null
    float* partial; int chunks_per_slice;   // split-K: blockIdx.z handles chunks [z*cps, (z+1)*cps) and writes partial[z]
    int epi_lds;             // the launch reserved enough LDS for the transposing epilogue (4 x 32 x (32 NB + 4) + 256 MB floats)
    // Folded 1x1x1 skip convolution (MyResBlock.skip_connection, diffusion_network.py:691,705): after the main chunks the
    // accumulators are rescaled (an exact power of two) and sk_cin more channels of the RAW tensors sk_in0 | sk_in1 are
    // accumulated through the centre tap only, with their own packed weights, input scale and bias.
    const float* sk_in0; const float* sk_in1;
    int sk_c0, sk_cin;
    const uint4* sk_w16; const float* sk_bias;
    const unsigned* sk_amax0; const unsigned* sk_amax1;
};

__device__ __forceinline__ int fast_div16(int n, int d, unsigned magic) {
    return (d == 1) ? n : (int)__umulhi((unsigned)n, magic);
}
__device__ __forceinline__ float act16(float t, int act) {
    if (act == 1) return t > 0.0f ? t : 0.02f * t;
    if (act == 2) return t / (1.0f + __expf(-t));
    return t;
}
// power-of-two scale that maps |x| <= bound to below 2^15 (fp16 max is 65504): s = 2^(14 - floor(log2 bound))
__device__ __forceinline__ int scale_exponent(float bound) {
    const unsigned bits = __float_as_uint(bound);
    const int eb = (int)((bits >> 23) & 0xffu) - 127;
    if (!(bound > 0.0f) || eb > 100) return 0;   // zero tensor, NaN or inf: any scale will do
    int e = 14 - eb;
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }

// One tiled body, two arithmetic variants.  256 threads = 4 waves; two workgroups share a CU (their staging and MFMA phases
// overlap by occupancy).  Rejected and removed after measurement (profiles/README.md r1w): a wave-specialised
// 512-thread variant (4 MFMA + 4 staging waves: 1.95 vs 1.54 ms) and a software-pipelined one-workgroup-per-CU variant
// (1.74-1.89 vs 1.59 ms); the timing-study switches that used to live in the tap loop are gone with them.
// EX ("exact"): the same tiling, staging, prologue and epilogue with fp32 operands on v_mfma_f32_32x32x2_f32 -- every product
// and every accumulation in fp32, as the reference's cuDNN/PyTorch fp32 convolution computes them (conv_precision = "f32").
// The LDS tile holds the 16 channels of a chunk as fp32 planes [channel][voxel] (the same 64 bytes per voxel as the four
// fp16 planes), a lane's B operand is one conflict-free ds_read_b32, its A operand one coalesced 4-byte load from the
// [tap][c_in][c_out] weight array (L2-resident: 442 KB for the 64 -> 64 layer), fetched one tap ahead.
// (Round 5 tried taking the dx = 1, 2 B fragments of a row from the neighbouring lane with v_mov_b32_dpp wave_shl:1 instead of two more
// LDS reads: 1.45 vs 1.30 ms on the dominant layer, 1.97 vs 1.78 J per launch -- profiles/r5c_conv_dppb_rejected.txt.)
template <int KS, int MB, int NB, bool EX = false>
__device__ __forceinline__ void conv3d_f16x3_body(const Conv16Args& A) {
    extern __shared__ uint4 smem16[];
    constexpr int PAD = (KS == 3) ? 1 : 0;
    constexpr int NW = 4, NT = 64 * NW;     // (an 8-wave, one-workgroup-per-CU tile was measured in round 4: 1.350 vs 1.323 ms, profiles/r4k_conv_eight_wave_tile_rejected.txt)
    const int bufsz = 4 * A.CS;            // one buffer: hi[2][CS], lo[2][CS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int kh = lane >> 5;
    const int l31 = lane & 31;

    // XCD-aware tile order: consecutive workgroups go to different XCDs (b % 8), so give each XCD a contiguous
    // run of tiles -- neighbouring tiles then share their halo planes in that XCD's L2.
    int t = blockIdx.x;
    {
        const int per = (A.n_tiles + 7) >> 3;
        const int cand = (t & 7) * per + (t >> 3);
        // exact only when n_tiles is a multiple of 8; otherwise keep the identity order
        if ((A.n_tiles & 7) == 0) t = cand;
    }
    const int tx = t % A.tiles_x; t /= A.tiles_x;
    const int ty = t % A.tiles_y;
    const int tz = t / A.tiles_y;
    const int ox0 = tx * A.TX, oy0 = ty * A.TY, oz0 = tz * A.TZ;
    const int cout0 = blockIdx.y * (MB * 32);
    const int lx0 = ox0 * A.stride - PAD, ly0 = oy0 * A.stride - PAD, lz0 = oz0 * A.stride - PAD;
    const size_t ISP = (size_t)A.ID * A.IH * A.IW;
    const size_t OSP = (size_t)A.OD * A.OH * A.OW;

    int voff[NB];
    int ovox[NB];
    bool valid[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = (wave * NB + nb) * 32 + l31;
        int x = j & (A.TX - 1);
        int y = (j >> A.lTX) & (A.TY - 1);
        int z = j >> (A.lTX + A.lTY);
        const bool v = (z < A.TZ) && (ox0 + x < A.OW) && (oy0 + y < A.OH) && (oz0 + z < A.OD);
        if (!v) { x = 0; y = 0; z = 0; }
        voff[nb] = ((z * A.HY + y) * A.HX + x) * A.stride + kh * A.CS;
        ovox[nb] = ((oz0 + z) * A.OH + (oy0 + y)) * A.OW + ox0 + x;
        valid[nb] = v;
    }

    // input scale
    float bound = A.in_bound;
    if (A.amax0) {
        bound = __uint_as_float(*A.amax0);
        if (A.amax1) bound = fmaxf(bound, __uint_as_float(*A.amax1));
    }
    const int ex = EX ? 0 : scale_exponent(bound);
    const float sx = pow2i(ex);

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    const int KG = A.cin >> 3;                              // 8-channel groups
    const size_t tap_stride = (size_t)KG * A.coutp;         // uint4 units
    const size_t plane = (size_t)(KS * KS * KS) * tap_stride;
    const uint4* wHi = A.w16 + kW16HeaderU4 + (size_t)kh * A.coutp + cout0 + l31;
    const uint4* wLo = wHi + plane;
    const float* wF = A.wf + (size_t)kh * A.coutp + cout0 + l31;     // EX: channel (2 kp + kh) of the pair, row l31 of block mb

    constexpr int TAPS = KS * KS * KS;
    // ---- stage the 16-channel chunk starting at c_base into `buf`: one voxel x 16 channels per item, two items per
    // thread in flight.  Everything uniform is kept out of the per-element code: the channel base pointers are scalar,
    // the 32 per-channel prologue constants are fetched with ONE vector load per wave and broadcast into SGPRs with
    // v_readlane, out-of-range voxels load element 0 (masked after the activation) so the 32 + 4 loads of an item pair
    // issue back to back without exec-mask branches, and the (has-prologue, activation) switch selects one of six
    // straight-line conversion bodies.  (The first version left those decisions to per-element code; the compiler
    // turned the constants into dependent vector loads with s_waitcnt vmcnt(0) -- 16 of the 27 us a chunk took.)
    auto stage_chunk = [&](int c_base, uint4* buf, int stid, int nthr) {
        uint4* bHi = buf;
        uint4* bLo = buf + 2 * A.CS;
        float pa[16], pb[16];
        {
            float pv = 0.0f;
            if (A.pro_a) pv = (lane & 16) ? A.pro_b[c_base + (lane & 15)] : A.pro_a[c_base + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                pa[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pv), j));
                pb[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pv), 16 + j));
            }
        }
        const bool has_aff = A.gamma != nullptr;
        for (int v0 = stid; v0 < A.CS; v0 += 2 * nthr) {
            float val[2][16];
            float gm[2], bt[2];
            int sidx[2];
            bool ok[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int vox = v0 + u * nthr;
                int rem = vox;
                const int hz = fast_div16(rem, A.HYX, A.mHYX);
                rem -= hz * A.HYX;
                const int hy = fast_div16(rem, A.HX, A.mHX);
                const int hx = rem - hy * A.HX;
                const int lz = lz0 + hz, ly = ly0 + hy, lx = lx0 + hx;
                ok[u] = vox < A.CS && (unsigned)lz < (unsigned)A.LD && (unsigned)ly < (unsigned)A.LH && (unsigned)lx < (unsigned)A.LW;
                sidx[u] = ok[u] ? ((lz >> A.ups) * A.IH + (ly >> A.ups)) * A.IW + (lx >> A.ups) : 0;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int cg = c_base + j;
                    const float* src = (cg < A.c0) ? (A.in0 + (size_t)cg * ISP) : (A.in1 + (size_t)(cg - A.c0) * ISP);
                    val[u][j] = src[sidx[u]];
                }
                gm[u] = 1.0f; bt[u] = 0.0f;
                if (has_aff) { gm[u] = A.gamma[sidx[u]]; bt[u] = A.beta[sidx[u]]; }
            }
            auto convert = [&](auto has_pro, auto act_c) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int vox = v0 + u * nthr;
                    if (EX) {
                        float* bF = reinterpret_cast<float*>(buf);
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            float t = val[u][j];
                            if (decltype(has_pro)::value) t = t * pa[j] + pb[j];
                            t = t * gm[u] + bt[u];
                            const float sc = ok[u] ? act16(t, decltype(act_c)::value) : 0.0f;   // zero padding AFTER the activation
                            if (vox < A.CS) bF[j * A.CS + vox] = sc;
                        }
                        continue;
                    }
                    f16x8 vh[2], vl[2];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        float t = val[u][j];
                        if (decltype(has_pro)::value) t = t * pa[j] + pb[j];
                        t = t * gm[u] + bt[u];
                        const float sc = ok[u] ? act16(t, decltype(act_c)::value) * sx : 0.0f;   // zero padding AFTER the activation
                        const _Float16 h = (_Float16)sc;
                        vh[j >> 3][j & 7] = h;
                        vl[j >> 3][j & 7] = (_Float16)(sc - (float)h);
                    }
                    if (vox < A.CS) {
                        bHi[vox] = __builtin_bit_cast(uint4, vh[0]);
                        bHi[A.CS + vox] = __builtin_bit_cast(uint4, vh[1]);
                        bLo[vox] = __builtin_bit_cast(uint4, vl[0]);
                        bLo[A.CS + vox] = __builtin_bit_cast(uint4, vl[1]);
                    }
                }
            };
            using T = std::true_type; using F = std::false_type;
            if (A.pro_a) {
                if (A.act == 1) convert(T{}, std::integral_constant<int, 1>{});
                else if (A.act == 2) convert(T{}, std::integral_constant<int, 2>{});
                else convert(T{}, std::integral_constant<int, 0>{});
            } else {
                if (A.act == 1) convert(F{}, std::integral_constant<int, 1>{});
                else if (A.act == 2) convert(F{}, std::integral_constant<int, 2>{});
                else convert(F{}, std::integral_constant<int, 0>{});
            }
        }
    };
    // ---- MFMA over the taps of one chunk; the A fragments of tap t+1 are fetched during tap t ----
    auto mfma_chunk = [&](int c_base, const uint4* buf) {
        __builtin_amdgcn_s_setprio(3);   // the MFMA phase issues ahead of a co-resident workgroup's staging (+1 % measured)
        if constexpr (EX) {
            const float* ldsF = reinterpret_cast<const float*>(buf);
            const size_t tapw = (size_t)A.cin * A.coutp;                 // floats per tap
            const float* wc = wF + (size_t)c_base * A.coutp;
            float a[8][MB], an[8][MB];
#pragma unroll
            for (int kp = 0; kp < 8; ++kp)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) a[kp][mb] = wc[(size_t)(2 * kp) * A.coutp + mb * 32];
#pragma unroll 1
            for (int zy = 0; zy < KS * KS; ++zy) {
                const int dz = zy / KS, dy = zy - dz * KS;
                const int rowoff = (dz * A.HY + dy) * A.HX;
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) {
                    const int tap = zy * KS + dx;
                    const int nxt = (tap + 1 < TAPS) ? tap + 1 : tap;
#pragma unroll
                    for (int kp = 0; kp < 8; ++kp)
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) an[kp][mb] = wc[(size_t)nxt * tapw + (size_t)(2 * kp) * A.coutp + mb * 32];
                    __builtin_amdgcn_sched_barrier(0);   // (as below: keep the next tap's A loads up here)
#pragma unroll
                    for (int kp = 0; kp < 8; ++kp) {
                        float b[NB];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) b[nb] = ldsF[voff[nb] + rowoff + dx + 2 * kp * A.CS];
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp][mb], b[nb], acc[mb][nb], 0, 0, 0);
                    }
#pragma unroll
                    for (int kp = 0; kp < 8; ++kp)
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) a[kp][mb] = an[kp][mb];
                }
            }
            __builtin_amdgcn_s_setprio(0);
            return;
        }
        const uint4* ldsHi = buf;
        const uint4* ldsLo = buf + 2 * A.CS;
        const uint4* wh = wHi + (size_t)(c_base >> 3) * A.coutp;
        const uint4* wl = wLo + (size_t)(c_base >> 3) * A.coutp;
        f16x8 ah[MB], al[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            ah[mb] = __builtin_bit_cast(f16x8, wh[mb * 32]);
            al[mb] = __builtin_bit_cast(f16x8, wl[mb * 32]);
        }
#pragma unroll 1
        for (int zy = 0; zy < KS * KS; ++zy) {
            const int dz = zy / KS, dy = zy - dz * KS;
            const int rowoff = (dz * A.HY + dy) * A.HX;
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
                const int tap = zy * KS + dx;
                const int nxt = (tap + 1 < TAPS) ? tap + 1 : tap;   // last tap: re-read itself (harmless)
                f16x8 ahn[MB], aln[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    ahn[mb] = __builtin_bit_cast(f16x8, wh[(size_t)nxt * tap_stride + mb * 32]);
                    aln[mb] = __builtin_bit_cast(f16x8, wl[(size_t)nxt * tap_stride + mb * 32]);
                }
                // keep the A loads of the next tap up here: left alone, the scheduler sinks them to just before their first
                // use and every tap then waits a full L2 round trip (measured: 31 us per chunk instead of 22; fetching two
                // taps ahead instead of one gains nothing more)
                __builtin_amdgcn_sched_barrier(0);
                f16x8 bh[NB], bl[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    bh[nb] = __builtin_bit_cast(f16x8, ldsHi[voff[nb] + rowoff + dx]);
                    bl[nb] = __builtin_bit_cast(f16x8, ldsLo[voff[nb] + rowoff + dx]);
                }
                // small terms first, then the leading term; every accumulator is revisited after MB*NB MFMAs
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mb], bh[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bl[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bh[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) { ah[mb] = ahn[mb]; al[mb] = aln[mb]; }
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    float inv2 = 0.0f;   // unscale factor after a folded skip convolution
    {
        const int c_begin = A.partial ? (int)blockIdx.z * A.chunks_per_slice * 16 : 0;
        const int c_end = A.partial ? min(A.cin, c_begin + A.chunks_per_slice * 16) : A.cin;
        for (int c_base = c_begin; c_base < c_end; c_base += 16) {
            __syncthreads();  // previous chunk fully consumed
            stage_chunk(c_base, smem16, tid, NT);
            __syncthreads();
            mfma_chunk(c_base, smem16);
        }
        if (!EX && A.sk_w16) {
            // ---- the block's 1x1x1 skip convolution, in the same accumulators (so that out = conv(h) + skip(x) leaves this
            // launch; the skip tensor is never written or re-read).  acc holds sum (w s_w)(x s_x); the skip products carry
            // (s_w' s_x') instead, so acc is first multiplied by (s_w' s_x') / (s_w s_x) -- a power of two, exact.
            float sb = __uint_as_float(*A.sk_amax0);
            if (A.sk_amax1) sb = fmaxf(sb, __uint_as_float(*A.sk_amax1));
            const int ex2 = scale_exponent(sb);
            const float sx2 = pow2i(ex2);
            const float inv_main = __uint_as_float(A.w16[0].x) * pow2i(-ex);
            inv2 = __uint_as_float(A.sk_w16[0].x) * pow2i(-ex2);
            const float ratio = inv_main / inv2;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][nb][r] *= ratio;
            const int KG2 = A.sk_cin >> 3;
            const uint4* w2Hi = A.sk_w16 + kW16HeaderU4 + (size_t)kh * A.coutp + cout0 + l31;
            const uint4* w2Lo = w2Hi + (size_t)KG2 * A.coutp;          // one tap: the lo plane follows the hi plane
            const int centre = (KS == 3) ? (A.HY + 1) * A.HX + 1 : 0;   // LDS offset of the tile's own voxels inside the halo'd tile
            const int tvox = A.TX * A.TY * A.TZ;
            for (int c2 = 0; c2 < A.sk_cin; c2 += 16) {
                f16x8 ah[MB], al[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    ah[mb] = __builtin_bit_cast(f16x8, w2Hi[(size_t)(c2 >> 3) * A.coutp + mb * 32]);
                    al[mb] = __builtin_bit_cast(f16x8, w2Lo[(size_t)(c2 >> 3) * A.coutp + mb * 32]);
                }
                __syncthreads();   // previous chunk fully consumed
                uint4* bHi = smem16;
                uint4* bLo = smem16 + 2 * A.CS;
                for (int j = tid; j < tvox; j += NT) {   // raw values, interior voxels only (one tap: no halo)
                    const int x = j & (A.TX - 1), y = (j >> A.lTX) & (A.TY - 1), z = j >> (A.lTX + A.lTY);
                    const bool okv = (ox0 + x < A.OW) && (oy0 + y < A.OH) && (oz0 + z < A.OD);
                    const size_t sidx = okv ? ((size_t)(oz0 + z) * A.IH + (oy0 + y)) * A.IW + ox0 + x : 0;
                    float val[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int cg = c2 + q;
                        const float* src = (cg < A.sk_c0) ? (A.sk_in0 + (size_t)cg * ISP) : (A.sk_in1 + (size_t)(cg - A.sk_c0) * ISP);
                        val[q] = src[sidx];
                    }
                    f16x8 vh[2], vl[2];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float sc = okv ? val[q] * sx2 : 0.0f;
                        const _Float16 hq = (_Float16)sc;
                        vh[q >> 3][q & 7] = hq;
                        vl[q >> 3][q & 7] = (_Float16)(sc - (float)hq);
                    }
                    const int slot = (z * A.HY + y) * A.HX + x + centre;
                    bHi[slot] = __builtin_bit_cast(uint4, vh[0]);
                    bHi[A.CS + slot] = __builtin_bit_cast(uint4, vh[1]);
                    bLo[slot] = __builtin_bit_cast(uint4, vl[0]);
                    bLo[A.CS + slot] = __builtin_bit_cast(uint4, vl[1]);
                }
                __syncthreads();
                f16x8 bh[NB], bl[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    bh[nb] = __builtin_bit_cast(f16x8, bHi[voff[nb] + centre]);
                    bl[nb] = __builtin_bit_cast(f16x8, bLo[voff[nb] + centre]);
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mb], bh[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bl[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bh[nb], acc[mb][nb], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: unscale, + bias (+ residual); C/D layout: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    // With A.stats the statistics the NEXT layer's LayerNorm/GroupNorm needs are taken here, while the values are in
    // registers: per output channel the sum and sum of squares over this workgroup's voxels (lane -> 32-lane DPP
    // reduction -> 4 waves through LDS) go to stats[tile][c_out_padded][2] with plain stores, and the tile's |x|max
    // to *out_amax; pixie_stats_finalize adds the tiles up in fp64.  That replaces one full read of the tensor.
    const float inv = EX ? 1.0f : (A.sk_w16 ? inv2 : __uint_as_float(A.w16[0].x) * pow2i(-ex));
    if (A.partial) {   // split-K slice: raw partial sums; bias, residual and statistics belong to splitk_reduce_kernel
        float* dst = A.partial + (size_t)blockIdx.z * A.cout * OSP;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cout0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (co < A.cout) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        if (valid[nb]) dst[(size_t)co * OSP + ovox[nb]] = acc[mb][nb][r] * inv;
                }
            }
        return;
    }
    float* red = reinterpret_cast<float*>(smem16);   // [4 waves][MB*32 rows][2], reused after the last chunk
    float wmax = 0.0f;
    // workgroup-uniform: every accumulator element of this tile is a real output
    const bool full = A.epi_lds && cout0 + MB * 32 <= A.cout && (A.TX & 3) == 0 && (A.OW & 3) == 0 && NW * NB * 32 <= A.TX * A.TY * A.TZ &&
                      ox0 + A.TX <= A.OW && oy0 + A.TY <= A.OH && oz0 + A.TZ <= A.OD;
    if (full) {
        __syncthreads();   // every wave is done with the activation tile
        // Interior tile (every tile of the 128^3 layers).  The accumulators go through LDS (the activation tile is dead)
        // so that each lane ends up with 4 x-consecutive voxels of one row: 16-byte residual loads and stores, 8x fewer
        // memory instructions, and a short rolled loop instead of 256 unrolled scalar stores.  (Written naively from the
        // MFMA register layout, every residual load carried an s_waitcnt vmcnt(0) that also drained the preceding
        // store: 256 serialised memory round trips, 42 us of a 229 us workgroup.)
        constexpr int LDO = NB * 32 + 4;                                  // row stride in floats
        float* ldsO = reinterpret_cast<float*>(smem16) + wave * (32 * LDO);   // this wave's 32 rows; no other wave touches it
        red = reinterpret_cast<float*>(smem16) + NW * 32 * LDO;
        constexpr int LPR = NB * 8;          // lanes per row (4 voxels each)
        constexpr int RPI = 32 / LPR;        // rows per half-wave per iteration
        const int rsub = l31 / LPR, colq = 4 * (l31 % LPR), nbq = colq >> 5;
        const int jq = (wave * NB + nbq) * 32 + (colq & 31);
        const int xq = jq & (A.TX - 1), yq = (jq >> A.lTX) & (A.TY - 1), zq = jq >> (A.lTX + A.lTY);
        const size_t ovq = ((size_t)(oz0 + zq) * A.OH + (oy0 + yq)) * A.OW + ox0 + xq;
        const bool has_res = A.residual != nullptr, has_stats = A.stats != nullptr;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) ldsO[((r & 3) + 8 * (r >> 2) + 4 * kh) * LDO + nb * 32 + l31] = acc[mb][nb][r];
#pragma unroll 4
            for (int it = 0; it < 16 / RPI; ++it) {
                const int rowl = (2 * it + kh) * RPI + rsub;
                const int co = cout0 + mb * 32 + rowl;
                const float4 a4 = *reinterpret_cast<const float4*>(ldsO + rowl * LDO + colq);
                const float bv = (A.bias ? A.bias[co] : 0.0f) + (A.sk_bias ? A.sk_bias[co] : 0.0f);
                const size_t o = (size_t)co * OSP + ovq;
                float4 v4;
                v4.x = a4.x * inv + bv; v4.y = a4.y * inv + bv; v4.z = a4.z * inv + bv; v4.w = a4.w * inv + bv;
                if (has_res) {
                    const float4 r4 = *reinterpret_cast<const float4*>(A.residual + o);
                    v4.x += r4.x; v4.y += r4.y; v4.z += r4.z; v4.w += r4.w;
                }
                *reinterpret_cast<float4*>(A.out + o) = v4;
                if (has_stats) {
                    float s1 = (v4.x + v4.y) + (v4.z + v4.w);
                    float s2 = (v4.x * v4.x + v4.y * v4.y) + (v4.z * v4.z + v4.w * v4.w);
                    wmax = fmaxf(fmaxf(wmax, fmaxf(fabsf(v4.x), fabsf(v4.y))), fmaxf(fabsf(v4.z), fabsf(v4.w)));
#pragma unroll
                    for (int off = LPR / 2; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
                    if ((l31 % LPR) == 0) { red[(wave * MB * 32 + mb * 32 + rowl) * 2] = s1; red[(wave * MB * 32 + mb * 32 + rowl) * 2 + 1] = s2; }
                }
            }
        }
    } else {
    if (A.stats) __syncthreads();                    // every wave is done with the activation tile
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const int co = cout0 + row;
            float s1 = 0.0f, s2 = 0.0f;
            if (co < A.cout) {
                const float bv = (A.bias ? A.bias[co] : 0.0f) + (A.sk_bias ? A.sk_bias[co] : 0.0f);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (valid[nb]) {
                        const size_t o = (size_t)co * OSP + ovox[nb];
                        float val = acc[mb][nb][r] * inv + bv;
                        if (A.residual) val += A.residual[o];
                        A.out[o] = val;
                        s1 += val; s2 += val * val; wmax = fmaxf(wmax, fabsf(val));
                    }
                }
            }
            if (A.stats) {
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
                if (l31 == 0) { red[(wave * MB * 32 + row) * 2] = s1; red[(wave * MB * 32 + row) * 2 + 1] = s2; }
            }
        }
    }
    }
    if (A.stats) {
        __syncthreads();
        if (tid < MB * 32 * 2) {
            float t = red[tid];
#pragma unroll
            for (int w = 1; w < NW; ++w) t += red[w * MB * 64 + tid];      // fixed order
            A.stats[((size_t)blockIdx.x * A.coutp + cout0) * 2 + tid] = t;
        }
        if (A.out_amax) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off, 64));
            if (lane == 0 && wmax > 0.0f) atomicMax(A.out_amax, __float_as_uint(wmax));
        }
    }
}

template <int KS, int MB, int NB>
__global__ __launch_bounds__(256, 2) void conv3d_f16x3_kernel(Conv16Args A) {
    conv3d_f16x3_body<KS, MB, NB>(A);
}
// conv_precision = "f32": the exact-fp32 variant of the same body (v_mfma_f32_32x32x2_f32)
template <int KS, int MB, int NB>
__global__ __launch_bounds__(256, 2) void conv3d_exact_kernel(Conv16Args A) {
    conv3d_f16x3_body<KS, MB, NB, true>(A);
}
// The dominant layer of the BASELINE network -- 64 -> 64 channels, 3^3, stride 1, on >= 128^3 voxels (the full-resolution
// level: 41 % of a scene's FLOPs at 128^3; the 64^3 level has the same channel counts and stays on the template) -- under its own symbol, so that `rocprofv3 --kernel-trace --stats` reports it as
// its own row instead of pooling it with the other shapes that share the <3,2,4> instantiation.  Same code, same results.
__global__ __launch_bounds__(256, 2) void conv3d_f16x3_c64_fullres_kernel(Conv16Args A) {
    conv3d_f16x3_body<3, 2, 4>(A);
}

// stats[tile][coutp][2] (fp32, from the conv epilogues) -> sums[c][2] (fp64), one workgroup per channel
__global__ __launch_bounds__(256) void stats_finalize_kernel(const float* __restrict__ stats, int n_tiles, int coutp, double* __restrict__ sums) {
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int t = threadIdx.x; t < n_tiles; t += 256) {
        const float2 v = *reinterpret_cast<const float2*>(stats + ((size_t)t * coutp + c) * 2);
        s1 += v.x; s2 += v.y;
    }
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
    __shared__ double r[8];
    if ((threadIdx.x & 63) == 0) { r[2 * (threadIdx.x >> 6)] = s1; r[2 * (threadIdx.x >> 6) + 1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) { sums[2 * c] = r[0] + r[2] + r[4] + r[6]; sums[2 * c + 1] = r[1] + r[3] + r[5] + r[7]; }
}

// out = sum_s partial[s] + bias (+ residual), slices added in a fixed order (deterministic split-K)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int slices, long n_elems, long osp,
                                                            const float* __restrict__ bias, const float* residual, float* out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_elems) return;
    float v = partial[i];
    for (int s = 1; s < slices; ++s) v += partial[(size_t)s * n_elems + i];
    if (bias) v += bias[i / osp];
    if (residual) v += residual[i];
    out[i] = v;
}

// |x|max of a tensor as float bits (non-negative floats order like unsigned integers); caller zeroes the slot
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ slot) {
    float m = 0.0f;
    const long stride = (long)gridDim.x * 256 * 4;
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if ((reinterpret_cast<size_t>(x) & 15) == 0) {
        for (; i + 3 < n; i += stride) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            m = fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        for (long k = i; k < n && k < i + 4; ++k) m = fmaxf(m, fabsf(x[k]));
    } else {
        for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[k]));
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(slot, __float_as_uint(m));
}

// (c_out, c_in, taps) fp32 -> header + hi planes + lo planes of [tap][c_in/8][c_out_padded] x (8 x fp16)
__global__ void pack_weights_f16x2_kernel(const float* __restrict__ src, uint4* __restrict__ dst, int cout, int cin, int taps,
                                          int coutp, const unsigned* __restrict__ amax_bits) {
    const int KG = cin >> 3;
    const long total = (long)taps * KG * coutp;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = scale_exponent(__uint_as_float(*amax_bits));
    if (i == 0) dst[0] = make_uint4(__float_as_uint(pow2i(-e)), (unsigned)cout, (unsigned)cin, (unsigned)taps);
    if (i >= total) return;
    const float s = pow2i(e);
    const int co = (int)(i % coutp);
    const long row = i / coutp;
    const int kg = (int)(row % KG);
    const int tap = (int)(row / KG);
    f16x8 vh, vl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float w = 0.0f;
        if (co < cout) w = src[((long)co * cin + kg * 8 + j) * taps + tap] * s;
        const _Float16 h = (_Float16)w;
        vh[j] = h;
        vl[j] = (_Float16)(w - (float)h);
    }
    dst[kW16HeaderU4 + i] = __builtin_bit_cast(uint4, vh);
    dst[kW16HeaderU4 + total + i] = __builtin_bit_cast(uint4, vl);
}

static unsigned magic_of16(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d); }
static int ilog2_16(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static int pow2_le16(int v, int cap) { int p = 1; while (p * 2 <= v && p * 2 <= cap) p *= 2; return p; }

template <int KS, int MB, int NB>
static int launch_f16x3(const Conv16Args& a, size_t lds_bytes, dim3 grid, hipStream_t st) {
    auto kern = conv3d_f16x3_kernel<KS, MB, NB>;
    PX_CHECK_HIP(allow_max_dynamic_lds(reinterpret_cast<const void*>(kern)));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, a);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

// geometry + tile selection shared by the launcher, pixie_conv_stats_floats and pixie_stats_finalize
static void conv16_tiling(const pixie_conv_desc* d, Conv16Args& a, int& MB_out, int& NB_out, int* slices_out = nullptr) {
    a.ID = d->in_d; a.IH = d->in_h; a.IW = d->in_w;
    a.ups = d->upsample;
    a.LD = a.ID << a.ups; a.LH = a.IH << a.ups; a.LW = a.IW << a.ups;
    const int pad = d->ksize == 3 ? 1 : 0;
    a.stride = d->stride;
    a.OD = (a.LD + 2 * pad - d->ksize) / a.stride + 1; a.OH = (a.LH + 2 * pad - d->ksize) / a.stride + 1; a.OW = (a.LW + 2 * pad - d->ksize) / a.stride + 1;
    if (d->out_d > 0) a.OD = std::min(a.OD, (int)d->out_d);   // odd-grid crop (diffusion_network.py:925-930)
    if (d->out_h > 0) a.OH = std::min(a.OH, (int)d->out_h);
    if (d->out_w > 0) a.OW = std::min(a.OW, (int)d->out_w);
    a.cout = d->c_out; a.coutp = pixie_conv_cout_padded(d->c_out);
    const long ovol = (long)a.OD * a.OH * a.OW;
    int MB = (a.coutp >= 64) ? 2 : 1;
    int NB = (a.stride == 2) ? 1 : 4;   // stride 2: the tile holds 8x the voxels it produces; only NB = 1 fits (112 KB, one workgroup per CU)
    auto n_wg = [&](int mb, int nb) {
        const long tiles = (ovol + 128L * nb - 1) / (128L * nb);
        return tiles * ((a.coutp + mb * 32 - 1) / (mb * 32));
    };
    // Split-K (needs the caller's workspace): when the output is too small to give every CU two workgroups, keep the
    // big MFMA-efficient tile and split the channel chunks over up to 8 slices instead of shrinking the tile; the
    // slices write partial outputs that splitk_reduce_kernel adds in a fixed order.
    int slices = 1;
    const int chunks = (d->c0 + d->c1) / 16;
    const int smax = d->d_workspace ? std::min(8, chunks / 2) : 1;
    if (smax >= 2 && n_wg(MB, NB) < 512) {
        bool found = false;
        for (int nb = NB; nb >= 1 && !found; nb /= 2) {
            for (int sl = 1; sl <= smax; sl *= 2)
                if (n_wg(MB, nb) * sl >= 512) { NB = nb; slices = sl; found = true; break; }
        }
        if (!found) { NB = 1; slices = 1; while (slices * 2 <= smax) slices *= 2; }
    } else {
        while (n_wg(MB, NB) < 512 && NB > 1) NB /= 2;
        if (n_wg(MB, NB) < 512 && MB > 1) MB = 1;
    }
    if (slices_out) *slices_out = slices;
    const int tile_vox = 128 * NB;
    a.TX = pow2_le16(a.OW, 32);
    a.TY = pow2_le16(a.OH, std::max(1, std::min(4, tile_vox / a.TX)));
    a.TZ = std::max(1, std::min(a.OD, tile_vox / (a.TX * a.TY)));
    a.lTX = ilog2_16(a.TX); a.lTY = ilog2_16(a.TY);
    a.tiles_x = (a.OW + a.TX - 1) / a.TX; a.tiles_y = (a.OH + a.TY - 1) / a.TY; a.tiles_z = (a.OD + a.TZ - 1) / a.TZ;
    a.n_tiles = a.tiles_x * a.tiles_y * a.tiles_z;
    a.HX = (a.TX - 1) * a.stride + d->ksize; a.HY = (a.TY - 1) * a.stride + d->ksize; a.HZ = (a.TZ - 1) * a.stride + d->ksize;
    a.HYX = a.HY * a.HX; a.CS = a.HZ * a.HYX;
    a.mHX = magic_of16(a.HX); a.mHYX = magic_of16(a.HYX);

    MB_out = MB; NB_out = NB;
}

// called by pixie_conv3d_forward (conv3d_mfma.hip) when the descriptor carries f16x2-packed weights
int conv3d_f16x3_forward(const pixie_conv_desc* d, hipStream_t st) {
    PX_REQUIRE(d->stride == 1 || (d->stride == 2 && d->ksize == 3 && !d->upsample), "f16x3 conv: stride must be 1, or 2 for a 3^3 kernel");
    const int cin = d->c0 + d->c1;
    PX_REQUIRE(cin % 16 == 0 && d->c0 % 8 == 0, "f16x3 conv: c_in must be a multiple of 16 (got %d+%d)", d->c0, d->c1);
    PX_REQUIRE(d->d_in_amax0 != nullptr || d->in_bound > 0.0f, "f16x3 conv: needs d_in_amax0 or a positive in_bound");
    PX_REQUIRE(d->c1 == 0 || d->d_in_amax0 == nullptr || d->d_in_amax1 != nullptr, "f16x3 conv: second input needs its own amax slot");
    Conv16Args a{};
    a.in0 = d->d_in0; a.in1 = d->d_in1; a.c0 = d->c0; a.cin = cin;
    a.ID = d->in_d; a.IH = d->in_h; a.IW = d->in_w;
    a.ups = d->upsample;
    a.LD = a.ID << a.ups; a.LH = a.IH << a.ups; a.LW = a.IW << a.ups;
    const int pad = d->ksize == 3 ? 1 : 0;
    a.stride = d->stride;
    a.OD = (a.LD + 2 * pad - d->ksize) / a.stride + 1; a.OH = (a.LH + 2 * pad - d->ksize) / a.stride + 1; a.OW = (a.LW + 2 * pad - d->ksize) / a.stride + 1;
    if (d->out_d > 0) a.OD = std::min(a.OD, (int)d->out_d);   // odd-grid crop (diffusion_network.py:925-930)
    if (d->out_h > 0) a.OH = std::min(a.OH, (int)d->out_h);
    if (d->out_w > 0) a.OW = std::min(a.OW, (int)d->out_w);
    a.pro_a = d->d_pro_a; a.pro_b = d->d_pro_b; a.gamma = d->d_gamma; a.beta = d->d_beta; a.act = d->act;
    a.w16 = reinterpret_cast<const uint4*>(d->d_w16); a.bias = d->d_bias; a.cout = d->c_out; a.coutp = pixie_conv_cout_padded(d->c_out);
    a.residual = d->d_residual; a.out = d->d_out;
    a.amax0 = d->d_in_amax0; a.amax1 = (d->c1 > 0) ? d->d_in_amax1 : nullptr; a.in_bound = d->in_bound;
    a.stats = d->d_out_stats; a.out_amax = d->d_out_amax;

    int MB = 0, NB = 0, slices = 1;
    conv16_tiling(d, a, MB, NB, &slices);
    if (d->d_skip_w16) {
        const int scin = d->skip_c0 + d->skip_c1;
        PX_REQUIRE(pixie_conv_skip_foldable(d), "f16x3 conv: this launch cannot fold a skip convolution (pixie_conv_skip_foldable)");
        PX_REQUIRE(d->d_skip_in0 && d->d_skip_amax0 && (d->skip_c1 == 0 || (d->d_skip_in1 && d->d_skip_amax1)), "f16x3 conv: folded skip needs its inputs and their amax slots");
        a.sk_in0 = d->d_skip_in0; a.sk_in1 = d->d_skip_in1; a.sk_c0 = d->skip_c0; a.sk_cin = scin;
        a.sk_w16 = reinterpret_cast<const uint4*>(d->d_skip_w16); a.sk_bias = d->d_skip_bias;
        a.sk_amax0 = d->d_skip_amax0; a.sk_amax1 = d->skip_c1 > 0 ? d->d_skip_amax1 : nullptr;
    }
    if (slices > 1) {
        a.partial = static_cast<float*>(d->d_workspace);
        a.chunks_per_slice = (cin / 16 + slices - 1) / slices;
        a.stats = nullptr; a.out_amax = nullptr;   // pixie_conv_stats_floats reports 0 for these layers
    }

    size_t lds = (size_t)4 * a.CS * sizeof(uint4);
    PX_REQUIRE(lds <= 160 * 1024, "f16x3 conv: tile needs %zu B of LDS", lds);
    {   // room for the transposing epilogue, as long as two workgroups still fit on a CU
        const size_t epi = ((size_t)4 * 32 * (NB * 32 + 4) + (size_t)4 * MB * 32 * 2) * sizeof(float);
        if (slices == 1 && epi <= 80 * 1024) { a.epi_lds = 1; if (lds < epi) lds = epi; }
    }
    const dim3 grid((unsigned)a.n_tiles, (unsigned)((a.coutp + MB * 32 - 1) / (MB * 32)), (unsigned)slices);
    if (slices > 1) {
        int rc = 1;
#define PX_CONV16_SK(KS_, MB_, NB_) \
        if (d->ksize == KS_ && MB == MB_ && NB == NB_) rc = launch_f16x3<KS_, MB_, NB_>(a, lds, grid, st);
        PX_CONV16_SK(3, 2, 4) PX_CONV16_SK(3, 2, 2) PX_CONV16_SK(3, 2, 1) PX_CONV16_SK(3, 1, 4) PX_CONV16_SK(3, 1, 2) PX_CONV16_SK(3, 1, 1)
        PX_CONV16_SK(1, 2, 4) PX_CONV16_SK(1, 2, 2) PX_CONV16_SK(1, 2, 1) PX_CONV16_SK(1, 1, 4) PX_CONV16_SK(1, 1, 2) PX_CONV16_SK(1, 1, 1)
#undef PX_CONV16_SK
        if (rc) return rc;
        const long osp = (long)a.OD * a.OH * a.OW, n_elems = (long)a.cout * osp;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n_elems + 255) / 256)), dim3(256), 0, st, a.partial, slices, n_elems, osp,
                           a.bias, a.residual, a.out);
        PX_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (d->ksize == 3 && MB == 2 && NB == 4 && cin == 64 && d->c_out == 64 && d->stride == 1 && !d->upsample && d->c1 == 0 && !a.sk_w16 &&
        (long)a.OD * a.OH * a.OW >= 128L * 128 * 128) {
        auto kern = conv3d_f16x3_c64_fullres_kernel;
        PX_CHECK_HIP(allow_max_dynamic_lds(reinterpret_cast<const void*>(kern)));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
        PX_CHECK_HIP(hipGetLastError());
        return 0;
    }
#define PX_CONV16_CASE(KS_, MB_, NB_) \
    if (d->ksize == KS_ && MB == MB_ && NB == NB_) return launch_f16x3<KS_, MB_, NB_>(a, lds, grid, st);
    PX_CONV16_CASE(3, 2, 4) PX_CONV16_CASE(3, 2, 2) PX_CONV16_CASE(3, 2, 1)
    PX_CONV16_CASE(3, 1, 4) PX_CONV16_CASE(3, 1, 2) PX_CONV16_CASE(3, 1, 1)
    PX_CONV16_CASE(1, 2, 4) PX_CONV16_CASE(1, 2, 2) PX_CONV16_CASE(1, 2, 1)
    PX_CONV16_CASE(1, 1, 4) PX_CONV16_CASE(1, 1, 2) PX_CONV16_CASE(1, 1, 1)
#undef PX_CONV16_CASE
    return set_error("f16x3 conv: no kernel variant for ksize=%d MB=%d NB=%d", d->ksize, MB, NB);
}

template <int KS, int MB, int NB>
static int launch_exact(const Conv16Args& a, size_t lds_bytes, dim3 grid, hipStream_t st) {
    auto kern = conv3d_exact_kernel<KS, MB, NB>;
    PX_CHECK_HIP(allow_max_dynamic_lds(reinterpret_cast<const void*>(kern)));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, a);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

// 1 when pixie_conv3d_forward sends an exact-fp32 descriptor (d_w, no d_w16) through the tiled body above instead of the
// first-generation kernel of conv3d_mfma.hip (odd channel counts, tiny test networks)
bool conv3d_exact_tiled_ok(const pixie_conv_desc* d) {
    const int cin = d->c0 + d->c1;
    return cin % 16 == 0 && (d->stride == 1 || (d->stride == 2 && d->ksize == 3 && !d->upsample)) && getenv("PIXIE_CONV_EXACT_V1") == nullptr;
}

// called by pixie_conv3d_forward (conv3d_mfma.hip) for exact-fp32 descriptors that conv3d_exact_tiled_ok accepts
int conv3d_exact_forward(const pixie_conv_desc* d, hipStream_t st) {
    Conv16Args a{};
    a.in0 = d->d_in0; a.in1 = d->d_in1; a.c0 = d->c0; a.cin = d->c0 + d->c1;
    a.pro_a = d->d_pro_a; a.pro_b = d->d_pro_b; a.gamma = d->d_gamma; a.beta = d->d_beta; a.act = d->act;
    a.wf = d->d_w; a.bias = d->d_bias;
    a.residual = d->d_residual; a.out = d->d_out;
    a.in_bound = 1.0f;
    pixie_conv_desc probe = *d;
    probe.d_workspace = nullptr;            // no split-K on this path: small layers shrink the tile instead
    int MB = 0, NB = 0, slices = 1;
    conv16_tiling(&probe, a, MB, NB, &slices);
    size_t lds = (size_t)4 * a.CS * sizeof(uint4);      // 16 fp32 planes = the four 16-byte fp16 planes
    PX_REQUIRE(lds <= 160 * 1024, "exact conv: tile needs %zu B of LDS", lds);
    {   // room for the transposing epilogue, as long as two workgroups still fit on a CU
        const size_t epi = ((size_t)4 * 32 * (NB * 32 + 4) + (size_t)4 * MB * 32 * 2) * sizeof(float);
        if (epi <= 80 * 1024) { a.epi_lds = 1; if (lds < epi) lds = epi; }
    }
    const dim3 grid((unsigned)a.n_tiles, (unsigned)((a.coutp + MB * 32 - 1) / (MB * 32)), 1u);
#define PX_CONVEX_CASE(KS_, MB_, NB_) \
    if (d->ksize == KS_ && MB == MB_ && NB == NB_) return launch_exact<KS_, MB_, NB_>(a, lds, grid, st);
    PX_CONVEX_CASE(3, 2, 4) PX_CONVEX_CASE(3, 2, 2) PX_CONVEX_CASE(3, 2, 1)
    PX_CONVEX_CASE(3, 1, 4) PX_CONVEX_CASE(3, 1, 2) PX_CONVEX_CASE(3, 1, 1)
    PX_CONVEX_CASE(1, 2, 4) PX_CONVEX_CASE(1, 2, 2) PX_CONVEX_CASE(1, 2, 1)
    PX_CONVEX_CASE(1, 1, 4) PX_CONVEX_CASE(1, 1, 2) PX_CONVEX_CASE(1, 1, 1)
#undef PX_CONVEX_CASE
    return set_error("exact conv: no kernel variant for ksize=%d MB=%d NB=%d", d->ksize, MB, NB);
}

}  // namespace pixie

using namespace pixie;

// number of floats of the epilogue statistics buffer for this descriptor (0 if the layer does not take the f16x3 path)
extern "C" int64_t pixie_conv_stats_floats(const pixie_conv_desc* d) {
    if (!d || !d->d_w16 || !(d->stride == 1 || (d->stride == 2 && d->ksize == 3)) || (d->ksize != 1 && d->ksize != 3)) return 0;
    Conv16Args a{};
    int MB = 0, NB = 0, slices = 1;
    conv16_tiling(d, a, MB, NB, &slices);
    return slices > 1 ? 0 : (int64_t)a.n_tiles * a.coutp * 2;
}

// bytes of d_workspace this layer can use for split-K (0: it would not split).  Decided on the shape alone.
extern "C" int64_t pixie_conv_workspace_bytes(const pixie_conv_desc* d) {
    if (!d || !d->d_w16 || !(d->stride == 1 || (d->stride == 2 && d->ksize == 3)) || (d->ksize != 1 && d->ksize != 3)) return 0;
    pixie_conv_desc probe = *d;
    probe.d_workspace = reinterpret_cast<void*>(1);   // "a workspace would be available"
    Conv16Args a{};
    int MB = 0, NB = 0, slices = 1;
    conv16_tiling(&probe, a, MB, NB, &slices);
    return slices > 1 ? (int64_t)slices * a.cout * a.OD * a.OH * a.OW * (int64_t)sizeof(float) : 0;
}

// which kernel instantiation pixie_conv3d_forward picks for this descriptor: variant = ksize * 100 + MB * 10 + NB of
// conv3d_f16x3_kernel<ksize, MB, NB>, slices = its split-K factor; 0 = the exact-fp32 kernel.  (Lets a profiler group
// its own per-launch timings the way rocprofv3 groups them: by kernel name.)
#ifdef PIXIE_DIAG
extern "C" int pixie_conv_kernel_variant(const pixie_conv_desc* d, int* slices_out) {
    if (slices_out) *slices_out = 1;
    if (!d || !d->d_w16 || !(d->stride == 1 || (d->stride == 2 && d->ksize == 3)) || (d->ksize != 1 && d->ksize != 3)) return 0;
    Conv16Args a{};
    int MB = 0, NB = 0, slices = 1;
    conv16_tiling(d, a, MB, NB, &slices);
    if (slices_out) *slices_out = slices;
    if (d->ksize == 3 && MB == 2 && NB == 4 && d->c0 + d->c1 == 64 && d->c_out == 64 && d->stride == 1 && !d->upsample && d->c1 == 0 && !d->d_skip_w16 &&
        (long)a.OD * a.OH * a.OW >= 128L * 128 * 128)
        return 9324;   // conv3d_f16x3_c64_fullres_kernel: the <3,2,4> code under its own symbol
    return d->ksize * 100 + MB * 10 + NB;
}
#endif

extern "C" int pixie_conv_skip_foldable(const pixie_conv_desc* d) {
    if (!d || !d->d_w16 || d->stride != 1 || d->upsample || (d->ksize != 1 && d->ksize != 3)) return 0;
    const int scin = d->skip_c0 + d->skip_c1;
    if (scin <= 0 || scin % 16 != 0 || d->skip_c0 % 8 != 0 || (d->c0 + d->c1) % 16 != 0 || d->c0 % 8 != 0) return 0;
    Conv16Args a{};
    int MB = 0, NB = 0, slices = 1;
    conv16_tiling(d, a, MB, NB, &slices);
    return slices == 1 ? 1 : 0;
}

extern "C" int pixie_stats_finalize(const float* d_stats, const pixie_conv_desc* d, double* d_sums, void* stream) {
    PX_REQUIRE(d_stats && d && d_sums, "pixie_stats_finalize: null argument");
    // recompute the tile count exactly as conv3d_f16x3_forward does
    Conv16Args a{};
    int MB = 0, NB = 0;
    conv16_tiling(d, a, MB, NB);
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((unsigned)d->c_out), dim3(256), 0, as_stream(stream), d_stats, a.n_tiles, a.coutp, d_sums);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int64_t pixie_conv_packed16_bytes(int c_out, int c_in, int ksize) {
    if (c_out <= 0 || c_in <= 0 || c_in % 8 != 0 || (ksize != 1 && ksize != 3)) return 0;
    const int64_t taps = (int64_t)ksize * ksize * ksize;
    return ((int64_t)kW16HeaderU4 + 2 * taps * (c_in / 8) * pixie_conv_cout_padded(c_out)) * (int64_t)sizeof(uint4);
}

extern "C" int pixie_tensor_amax(const float* d_x, int64_t count, uint32_t* d_slot, void* stream) {
    PX_REQUIRE(d_x && d_slot && count > 0, "pixie_tensor_amax: bad arguments");
    const int blocks = (int)std::min<int64_t>(2048, (count + 1023) / 1024);
    hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), d_x, (long)count, d_slot);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int pixie_conv_pack_weights_f16x2(const float* d_w, void* d_packed, int c_out, int c_in, int ksize, void* stream) {
    PX_REQUIRE(d_w && d_packed && c_out > 0 && c_in > 0 && c_in % 8 == 0 && (ksize == 1 || ksize == 3),
               "pixie_conv_pack_weights_f16x2: bad arguments (c_in must be a multiple of 8)");
    hipStream_t st = as_stream(stream);
    const int taps = ksize * ksize * ksize;
    const int coutp = pixie_conv_cout_padded(c_out);
    // the |w|max slot lives in the header's last word until the pack kernel overwrites the header
    unsigned* slot = reinterpret_cast<unsigned*>(d_packed) + 15;
    PX_CHECK_HIP(hipMemsetAsync(d_packed, 0, kW16HeaderU4 * sizeof(uint4), st));
    if (pixie_tensor_amax(d_w, (int64_t)c_out * c_in * taps, slot, stream)) return 1;
    const long total = (long)taps * (c_in / 8) * coutp;
    hipLaunchKernelGGL(pack_weights_f16x2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, d_w, reinterpret_cast<uint4*>(d_packed), c_out,
                       c_in, taps, coutp, slot);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}
