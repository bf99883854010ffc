// pixie_amd/csrc/conv3d_mfma.hip -- 3D convolution for the U-Net on gfx950 fp32 matrix cores.
//
// Replaces every nn.Conv3d / nn.Conv1d the reference graph launches through cuDNN
// (WG/models/module/diffusion_network.py: conv_nd at :679,:683,:691,:762,:58,:90,:206,:208,:872 and
// FeatureProjector :570-583) together with the op that precedes it in the graph: spatial LayerNorm +
// LeakyReLU (:674-676), GroupNorm (+SiLU) (:571-584, :199), nearest x2 upsampling (:69), th.cat (:932)
// and the residual add (:705).  None of those intermediates is materialised.
//
// Formulation (activations NCDHW fp32, batch 1):  Out[co][v] = sum_{tap,ci} W[tap][ci][co] * X[ci][v+tap]
// is an implicit GEMM with M = c_out, N = voxels, K = 27*c_in, computed with
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate: 64 cycles/SIMD each, so the matrix
// pipe -- not LDS or HBM -- is the bound by construction; arithmetic intensity of the 64->64 conv is
// ~430 FLOP/B).  A = weights (rows = c_out), B = activations (columns = 32 voxels that are contiguous
// in x), so both the LDS reads of B and the global stores of the accumulator are unit-stride.
//
// Workgroup = 4 waves.  Per chunk of CK input channels it stages into LDS
//   X tile  [CK][HZ][HY][HX]   the halo'd voxel tile, with the prologue (norm affine + activation)
//                              applied on the way in and zero padding applied AFTER it,
//   W tile  [k^3][CK][MB*32]   the weight slab for this c_out tile,
// then each wave runs k^3 * CK/2 steps of (MB + NB) ds_read_b32 + MB*NB MFMAs on its
// MB*32 (c_out) x NB*32 (voxels) accumulator block.
#include <hip/hip_runtime.h>

#include "../../include/pixie_hip.h"
#include "common.h"

namespace pixie {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* in0; const float* in1;
    int c0, cin;
    int ID, IH, IW;          // stored input dims
    int LD, LH, LW;          // logical input dims (after optional nearest x2)
    int ups;                 // 0 or 1: logical -> stored coordinate shift
    int stride;
    int OD, OH, OW;
    const float* pro_a; const float* pro_b; const float* gamma; const float* beta;
    int act;
    const float* w; const float* bias;
    int cout, coutp;
    const float* residual; float* out;
    // tiling (host-chosen): output tile TX x TY x TZ, TX/TY powers of two
    int TX, TY, TZ, lTX, lTY;
    int tiles_x, tiles_y, tiles_z;
    int HX, HY, HZ, HYX, CS;     // halo tile dims; HYX = HY*HX; CS = HZ*HY*HX
    unsigned mHX, mHYX, mCS;     // ceil(2^32/d) magic multipliers (0 when d == 1)
};

__device__ __forceinline__ int fast_div(int n, int d, unsigned magic) {
    return (d == 1) ? n : (int)__umulhi((unsigned)n, magic);
}

__device__ __forceinline__ float apply_act(float t, int act) {
    if (act == 1) return t > 0.0f ? t : 0.02f * t;              // LeakyReLU(0.02)
    if (act == 2) return t / (1.0f + __expf(-t));                // SiLU = x * sigmoid(x)
    return t;
}

template <int KS, int MB, int NB, int CK>
__global__ __launch_bounds__(256, 2) void conv3d_mfma_kernel(ConvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int MBW = MB * 32;
    constexpr int TAPS = KS * KS * KS;
    constexpr int PAD = (KS == 3) ? 1 : 0;
    float* ldsX = smem;
    float* ldsW = smem + ((CK * A.CS + 3) & ~3);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    int t = blockIdx.x;
    const int tx = t % A.tiles_x; t /= A.tiles_x;
    const int ty = t % A.tiles_y;
    const int tz = t / A.tiles_y;
    const int ox0 = tx * A.TX, oy0 = ty * A.TY, oz0 = tz * A.TZ;
    const int cout0 = blockIdx.y * MBW;
    // logical-input coordinate of halo element (0,0,0)
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
        voff[nb] = ((z * A.stride) * A.HY + y * A.stride) * A.HX + x * A.stride + half * A.CS;
        ovox[nb] = ((oz0 + z) * A.OH + (oy0 + y)) * A.OW + ox0 + x;
        valid[nb] = v;
    }
    const int laneA = half * MBW + l31;

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    const int tileX = CK * A.CS;
    for (int c_base = 0; c_base < A.cin; c_base += CK) {
        __syncthreads();  // previous chunk fully consumed
        // ---- stage the activation tile (prologue fused; zero padding after the activation) ----
        for (int idx = tid; idx < tileX; idx += 256) {
            const int c = fast_div(idx, A.CS, A.mCS);
            int rem = idx - c * A.CS;
            const int hz = fast_div(rem, A.HYX, A.mHYX);
            rem -= hz * A.HYX;
            const int hy = fast_div(rem, A.HX, A.mHX);
            const int hx = rem - hy * A.HX;
            const int lz = lz0 + hz, ly = ly0 + hy, lx = lx0 + hx;
            const int cg = c_base + c;
            float val = 0.0f;
            if (cg < A.cin && (unsigned)lz < (unsigned)A.LD && (unsigned)ly < (unsigned)A.LH && (unsigned)lx < (unsigned)A.LW) {
                const int sidx = ((lz >> A.ups) * A.IH + (ly >> A.ups)) * A.IW + (lx >> A.ups);
                const float* src = (cg < A.c0) ? (A.in0 + (size_t)cg * ISP) : (A.in1 + (size_t)(cg - A.c0) * ISP);
                val = src[sidx];
                if (A.pro_a) val = val * A.pro_a[cg] + A.pro_b[cg];
                if (A.gamma) val = val * A.gamma[sidx] + A.beta[sidx];
                val = apply_act(val, A.act);
            }
            ldsX[idx] = val;
        }
        // ---- stage the weight slab: rows (tap, c) of MBW contiguous c_out values ----
        constexpr int WE4 = TAPS * CK * MBW / 4;
        for (int i4 = tid; i4 < WE4; i4 += 256) {
            const int idx = i4 * 4;
            const int m = idx & (MBW - 1);
            const int row = idx / MBW;
            const int c = row % CK;
            const int tap = row / CK;
            const int cg = c_base + c;
            const int co = cout0 + m;
            float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cg < A.cin && co < A.coutp)
                wv = *reinterpret_cast<const float4*>(A.w + ((size_t)tap * A.cin + cg) * A.coutp + co);
            *reinterpret_cast<float4*>(ldsW + idx) = wv;
        }
        __syncthreads();
        // ---- MFMA over taps x channel pairs ----
#pragma unroll 1
        for (int dz = 0; dz < KS; ++dz) {
#pragma unroll 1
            for (int dy = 0; dy < KS; ++dy) {
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) {
                    const int tap = (dz * KS + dy) * KS + dx;
                    const int tapoff = (dz * A.HY + dy) * A.HX + dx;
#pragma unroll
                    for (int kp = 0; kp < CK / 2; ++kp) {
                        float a[MB], b[NB];
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) a[mb] = ldsW[(tap * CK + 2 * kp) * MBW + mb * 32 + laneA];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) b[nb] = ldsX[voff[nb] + 2 * kp * A.CS + tapoff];
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], b[nb], acc[mb][nb], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: + bias (+ residual), unit-stride stores along x ----
    // C/D layout of 32x32 MFMA: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cout0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < A.cout) {
                const float bv = A.bias ? A.bias[co] : 0.0f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (valid[nb]) {
                        const size_t o = (size_t)co * OSP + ovox[nb];
                        float val = acc[mb][nb][r] + bv;
                        if (A.residual) val += A.residual[o];
                        A.out[o] = val;
                    }
                }
            }
        }
    }
}

// (c_out, c_in, k,k,k) -> [tap][c_in][c_out_padded]
__global__ void pack_weights_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin, int taps, int coutp) {
    const long total = (long)taps * cin * coutp;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % coutp);
    const long row = i / coutp;
    const int ci = (int)(row % cin);
    const int tap = (int)(row / cin);
    dst[i] = (co < cout) ? src[((long)co * cin + ci) * taps + tap] : 0.0f;
}

static unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d); }
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static int pow2_le(int v, int cap) { int p = 1; while (p * 2 <= v && p * 2 <= cap) p *= 2; return p; }

template <int KS, int MB, int NB, int CK>
static int launch_variant(const ConvArgs& a, size_t lds_bytes, dim3 grid, hipStream_t st) {
    auto kern = conv3d_mfma_kernel<KS, MB, NB, CK>;
    PX_CHECK_HIP(allow_max_dynamic_lds(reinterpret_cast<const void*>(kern)));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, a);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace pixie

namespace pixie {
int conv3d_f16x3_forward(const pixie_conv_desc* d, hipStream_t st);
bool conv3d_exact_tiled_ok(const pixie_conv_desc* d);
int conv3d_exact_forward(const pixie_conv_desc* d, hipStream_t st);
}
using namespace pixie;

extern "C" int pixie_conv_cout_padded(int c_out) { return (c_out + 31) / 32 * 32; }

extern "C" int pixie_conv_pack_weights(const float* d_w, float* d_packed, int c_out, int c_in, int ksize, void* stream) {
    PX_REQUIRE(d_w && d_packed && c_out > 0 && c_in > 0 && (ksize == 1 || ksize == 3), "pixie_conv_pack_weights: bad arguments");
    const int taps = ksize * ksize * ksize;
    const int coutp = pixie_conv_cout_padded(c_out);
    const long total = (long)taps * c_in * coutp;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), d_w, d_packed, c_out, c_in, taps, coutp);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int pixie_conv3d_forward(const pixie_conv_desc* d, void* stream) {
    PX_REQUIRE(d && d->d_in0 && (d->d_w || d->d_w16) && d->d_out, "pixie_conv3d_forward: null pointer in descriptor");
    PX_REQUIRE(d->ksize == 1 || d->ksize == 3, "pixie_conv3d_forward: ksize must be 1 or 3 (got %d)", d->ksize);
    PX_REQUIRE(d->stride == 1 || d->stride == 2, "pixie_conv3d_forward: stride must be 1 or 2 (got %d)", d->stride);
    PX_REQUIRE(d->upsample == 0 || d->upsample == 1, "pixie_conv3d_forward: upsample must be 0 or 1");
    PX_REQUIRE(d->c0 > 0 && d->c1 >= 0 && (d->c1 == 0 || d->d_in1), "pixie_conv3d_forward: bad channel split");
    PX_REQUIRE(d->c_out > 0 && d->in_d > 0 && d->in_h > 0 && d->in_w > 0, "pixie_conv3d_forward: bad sizes");
    PX_REQUIRE((d->d_pro_a == nullptr) == (d->d_pro_b == nullptr), "pixie_conv3d_forward: pro_a/pro_b must come together");
    PX_REQUIRE((d->d_gamma == nullptr) == (d->d_beta == nullptr), "pixie_conv3d_forward: gamma/beta must come together");
    PX_REQUIRE(!(d->d_gamma && d->upsample), "pixie_conv3d_forward: spatial affine with upsample is not in the reference graph");

    if (d->d_w16) return conv3d_f16x3_forward(d, as_stream(stream));
    PX_REQUIRE(!d->d_skip_w16, "pixie_conv3d_forward: a folded skip convolution needs the f16x3 path (d_w16)");
    // 16-aligned channel counts (every layer of the reference networks): the tiled body of conv3d_f16x3.hip with fp32
    // operands; the kernel below keeps the odd shapes (tiny test networks, c_in = 3 ...)
    if (conv3d_exact_tiled_ok(d)) return conv3d_exact_forward(d, as_stream(stream));

    ConvArgs a{};
    a.in0 = d->d_in0; a.in1 = d->d_in1; a.c0 = d->c0; a.cin = d->c0 + d->c1;
    a.ID = d->in_d; a.IH = d->in_h; a.IW = d->in_w;
    a.ups = d->upsample;
    a.LD = a.ID << a.ups; a.LH = a.IH << a.ups; a.LW = a.IW << a.ups;
    a.stride = d->stride;
    const int pad = d->ksize == 3 ? 1 : 0;
    a.OD = (a.LD + 2 * pad - d->ksize) / a.stride + 1;
    a.OH = (a.LH + 2 * pad - d->ksize) / a.stride + 1;
    a.OW = (a.LW + 2 * pad - d->ksize) / a.stride + 1;
    if (d->out_d > 0) a.OD = std::min(a.OD, (int)d->out_d);   // odd-grid crop (diffusion_network.py:925-930)
    if (d->out_h > 0) a.OH = std::min(a.OH, (int)d->out_h);
    if (d->out_w > 0) a.OW = std::min(a.OW, (int)d->out_w);
    a.pro_a = d->d_pro_a; a.pro_b = d->d_pro_b; a.gamma = d->d_gamma; a.beta = d->d_beta; a.act = d->act;
    a.w = d->d_w; a.bias = d->d_bias; a.cout = d->c_out; a.coutp = pixie_conv_cout_padded(d->c_out);
    a.residual = d->d_residual; a.out = d->d_out;

    // ---- choose the tile: MB c_out blocks x (4 waves * NB) voxel blocks per workgroup ----
    const long ovol = (long)a.OD * a.OH * a.OW;
    int MB = (a.coutp >= 64) ? 2 : 1;
    int NB = 4;
    auto n_wg = [&](int mb, int nb) {
        const long tiles = (ovol + 128L * nb - 1) / (128L * nb);
        return tiles * ((a.coutp + mb * 32 - 1) / (mb * 32));
    };
    // keep at least ~2 workgroups per CU when the layer allows it (256 CUs)
    while (n_wg(MB, NB) < 512 && NB > 1) NB /= 2;
    if (n_wg(MB, NB) < 512 && MB > 1) MB = 1;
    if (a.stride == 2 && NB > 2) NB = 2;  // halo of a strided tile is ~8x larger; keep LDS in budget

    const int tile_vox = 128 * NB;
    a.TX = pow2_le(a.OW, 32);
    a.TY = pow2_le(a.OH, std::max(1, tile_vox / a.TX));
    a.TZ = std::max(1, std::min(a.OD, tile_vox / (a.TX * a.TY)));
    a.lTX = ilog2(a.TX); a.lTY = ilog2(a.TY);
    a.tiles_x = (a.OW + a.TX - 1) / a.TX; a.tiles_y = (a.OH + a.TY - 1) / a.TY; a.tiles_z = (a.OD + a.TZ - 1) / a.TZ;
    a.HX = (a.TX - 1) * a.stride + d->ksize; a.HY = (a.TY - 1) * a.stride + d->ksize; a.HZ = (a.TZ - 1) * a.stride + d->ksize;
    a.HYX = a.HY * a.HX; a.CS = a.HZ * a.HYX;
    a.mHX = magic_of(a.HX); a.mHYX = magic_of(a.HYX); a.mCS = magic_of(a.CS);

    const int CK = d->ksize == 3 ? 4 : 16;
    const int taps = d->ksize * d->ksize * d->ksize;
    const size_t lds = ((size_t)((CK * a.CS + 3) & ~3) + (size_t)taps * CK * MB * 32) * sizeof(float);
    PX_REQUIRE(lds <= 160 * 1024, "pixie_conv3d_forward: tile needs %zu B of LDS", lds);
    const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.tiles_z), (unsigned)((a.coutp + MB * 32 - 1) / (MB * 32)));
    hipStream_t st = as_stream(stream);

#define PX_CONV_CASE(KS_, MB_, NB_, CK_) \
    if (d->ksize == KS_ && MB == MB_ && NB == NB_) return launch_variant<KS_, MB_, NB_, CK_>(a, lds, grid, st);
    PX_CONV_CASE(3, 2, 4, 4) PX_CONV_CASE(3, 2, 2, 4) PX_CONV_CASE(3, 2, 1, 4)
    PX_CONV_CASE(3, 1, 4, 4) PX_CONV_CASE(3, 1, 2, 4) PX_CONV_CASE(3, 1, 1, 4)
    PX_CONV_CASE(1, 2, 4, 16) PX_CONV_CASE(1, 2, 2, 16) PX_CONV_CASE(1, 2, 1, 16)
    PX_CONV_CASE(1, 1, 4, 16) PX_CONV_CASE(1, 1, 2, 16) PX_CONV_CASE(1, 1, 1, 16)
#undef PX_CONV_CASE
    return set_error("pixie_conv3d_forward: no kernel variant for ksize=%d MB=%d NB=%d", d->ksize, MB, NB);
}
