// pixie_amd/csrc/unet_ops.hip -- the non-convolution operators of the U-Net path on gfx950.
//
//  channel sums + finalize   statistics of nn.LayerNorm([D,H,W]) (diffusion_network.py:674,679,870) and
//                            nn.GroupNorm / GroupNorm32 (:571-584; nn.py:17-19, :199) reduced to the
//                            per-channel (a,b) that the conv kernel's prologue applies, so that no
//                            normalised tensor is ever written to HBM;
//  attention                 QKVAttention.forward (diffusion_network.py:224-242), streamed softmax on the fp32 matrix cores;
//  combine                   argmax + one-hot + concat of inference_combined.py:124-126,186-195.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/pixie_hip.h"
#include "common.h"

namespace pixie {

// ---------------------------------------------------------------- per-channel sum / sum of squares
// grid = (splits, channels); each block reduces one contiguous segment of one channel with float4
// loads (HBM-bound: one read of the tensor), fp32 per-thread partials over <= a few hundred
// elements, fp64 from the block reduction on, one fp64 atomic pair per block.
__global__ __launch_bounds__(256) void channel_sums_kernel(const float* __restrict__ x, long spatial, long seg, double* __restrict__ sums,
                                                           unsigned* __restrict__ amax) {
    const int c = blockIdx.y;
    const long begin = (long)blockIdx.x * seg;
    const long end = begin + seg < spatial ? begin + seg : spatial;
    const float* p = x + (size_t)c * spatial;
    double s1 = 0.0, s2 = 0.0;
    float mx = 0.0f;
    const bool vec_ok = ((reinterpret_cast<size_t>(p + begin) & 15) == 0);
    long i = begin + (long)threadIdx.x * 4;
    if (vec_ok) {
        float a1 = 0.f, a2 = 0.f;
        int cnt = 0;
        for (; i + 3 < end; i += 256 * 4) {
            const float4 v = *reinterpret_cast<const float4*>(p + i);
            a1 += (v.x + v.y) + (v.z + v.w);
            a2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            mx = fmaxf(fmaxf(mx, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
            if (++cnt == 64) { s1 += a1; s2 += a2; a1 = 0.f; a2 = 0.f; cnt = 0; }
        }
        s1 += a1; s2 += a2;
        for (long k = i; k < end && k < i + 4; ++k) { const float v = p[k]; s1 += v; s2 += (double)v * v; mx = fmaxf(mx, fabsf(v)); }
    } else {
        for (long k = begin + threadIdx.x; k < end; k += 256) { const float v = p[k]; s1 += v; s2 += (double)v * v; mx = fmaxf(mx, fabsf(v)); }
    }
    // wave reduce (64 lanes) then across the 4 waves through LDS
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off, 64);
        s2 += __shfl_down(s2, off, 64);
        mx = fmaxf(mx, __shfl_down(mx, off, 64));
    }
    if (amax && (threadIdx.x & 63) == 0 && mx > 0.0f) atomicMax(amax, __float_as_uint(mx));
    __shared__ double red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[2 * wave] = s1; red[2 * wave + 1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t1 = red[0] + red[2] + red[4] + red[6];
        const double t2 = red[1] + red[3] + red[5] + red[7];
        atomicAdd(&sums[2 * c], t1);
        atomicAdd(&sums[2 * c + 1], t2);
    }
}

// mode 0: per-channel LayerNorm statistics (biased variance, eps inside the sqrt -- torch semantics)
// mode 1: GroupNorm(groups): statistics over (channels/groups) x spatial, affine folded per channel
// `sums` holds the first c0 channels, `sums1` (may be null) the channels from c0 on: the statistics of th.cat([h, skip]) are the
// concatenation of the two tensors' statistics, read where they lie
__global__ void norm_finalize_kernel(const double* __restrict__ sums, int c0, const double* __restrict__ sums1, int channels, double spatial,
                                     int mode, int groups, double eps, const float* __restrict__ weight, const float* __restrict__ bias,
                                     float* __restrict__ a, float* __restrict__ b) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= channels) return;
    auto S = [&](int k, int which) { return k < c0 ? sums[2 * k + which] : sums1[2 * (k - c0) + which]; };
    double mean, var;
    if (mode == 0) {
        mean = S(c, 0) / spatial;
        var = S(c, 1) / spatial - mean * mean;
    } else {
        const int cpg = channels / groups;
        const int g = c / cpg;
        double s1 = 0.0, s2 = 0.0;
        for (int k = g * cpg; k < (g + 1) * cpg; ++k) { s1 += S(k, 0); s2 += S(k, 1); }
        const double cnt = spatial * cpg;
        mean = s1 / cnt;
        var = s2 / cnt - mean * mean;
    }
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + eps);
    if (mode == 0) {
        a[c] = (float)rstd;
        b[c] = (float)(-mean * rstd);
    } else {
        const double w = weight ? (double)weight[c] : 1.0;
        const double bb = bias ? (double)bias[c] : 0.0;
        a[c] = (float)(rstd * w);
        b[c] = (float)(bb - mean * rstd * w);
    }
}

__global__ void channel_affine_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                      float* __restrict__ y, long spatial, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i / spatial);
    y[i] = x[i] * a[c] + b[c];
}

// ---------------------------------------------------------------- attention
// qkv is [3C][T] (q rows 0..C-1, k rows C..2C-1, v rows 2C..3C-1); out[c][t] = sum_s softmax_s(q_t . k_s * C^-1/2) v[c][s]
// (QKVAttention.forward, diffusion_network.py:224-242; one head of width C).  No T x T logits exist anywhere
// (the reference materialises them: 64 MiB at 128^3, 4 GiB at 256^3).
//
// Exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32).  One workgroup = 16 queries x all keys, 8 waves; wave w owns the
// key tiles {2(w+8j), 2(w+8j)+1} (pairs, so both halves of each 128-byte line of V are used by the same wave) and keeps
// its own streamed-softmax state (m, l, O^T); the eight states are merged once at the end in a fixed order.
//   S^T[key][q] = sum_c K[c][key] Q[c][q]   A = K tile straight from global (64-byte segments), B = Q from LDS
//   O^T[c][q]  += sum_s V[c][s] P[s][q]     A = V tile straight from global (float4 = the 4 keys a lane owns),
//                                           B = P = the S^T accumulator registers themselves (D layout == B layout)
// so neither P nor V passes through LDS, and the rescale factor is a per-lane scalar (queries are MFMA columns in both).
typedef float attn_f4 __attribute__((ext_vector_type(4)));
constexpr int ATT_BQ = 16;
constexpr int ATT_WAVES = 8;
template <int C>
__global__ __launch_bounds__(ATT_WAVES * 64) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int KS = C / 4;   // contraction steps of S^T
    constexpr int MT = C / 16;  // 16-channel row tiles of O^T
    float* Qs = sm;             // [C][16], pre-scaled; the region is reused for the merge
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, quad = lane >> 4;
    const int t0 = blockIdx.x * ATT_BQ;
    const float* q = qkv;
    const float* k = qkv + (size_t)C * T;
    const float* v = qkv + (size_t)2 * C * T;
    const float scale2 = 1.0f / sqrtf((float)C);  // (C^-1/4)^2, diffusion_network.py:233-236
    for (int i = tid; i < C * ATT_BQ; i += ATT_WAVES * 64) {
        const int c = i >> 4, qi = i & 15;
        Qs[i] = (t0 + qi < T) ? q[(size_t)c * T + t0 + qi] * scale2 : 0.0f;
    }
    __syncthreads();

    attn_f4 O[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) O[i] = attn_f4{0.f, 0.f, 0.f, 0.f};
    float m = -3.0e38f, l = 0.0f;
    const int ntiles = (T + 15) >> 4;
    const bool vec_ok = (T & 3) == 0;

    // Operand registers are loaded in halves so that at most ~160 of them are live: K[0:KS/2) of the next tile during the
    // PV products, K[KS/2:KS) and V[0:MT/2) at the start of S^T, V[MT/2:MT) at its midpoint.
    constexpr int KH = KS / 2, MH = MT / 2;
    float ka[KH], kb[KH];
    attn_f4 va[MH], vb[MH];
    auto load_k = [&](float (&dst)[KH], int tile, int kk0) {
        int key = tile * 16 + col;
        key = key < T ? key : T - 1;  // masked after the product
        const float* kp = k + (size_t)(4 * kk0 + quad) * T + key;
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) dst[kk] = kp[(size_t)(4 * kk) * T];
    };
    auto load_v = [&](attn_f4 (&dst)[MH], int tile, int mt0) {
        const int key = tile * 16 + 4 * quad;
        if (vec_ok && key + 3 < T) {
            const float* vp = v + (size_t)(16 * mt0 + col) * T + key;
#pragma unroll
            for (int mt = 0; mt < MH; ++mt) dst[mt] = *reinterpret_cast<const attn_f4*>(vp + (size_t)(16 * mt) * T);
        } else {
#pragma unroll
            for (int mt = 0; mt < MH; ++mt) {
                const float* vp = v + (size_t)(16 * (mt0 + mt) + col) * T;
                attn_f4 r;
                r.x = key + 0 < T ? vp[key + 0] : 0.0f;
                r.y = key + 1 < T ? vp[key + 1] : 0.0f;
                r.z = key + 2 < T ? vp[key + 2] : 0.0f;
                r.w = key + 3 < T ? vp[key + 3] : 0.0f;
                dst[mt] = r;
            }
        }
    };
    auto tile_of = [&](int it) { return (((it >> 1) * ATT_WAVES + wave) << 1) + (it & 1); };
    auto pv = [&](attn_f4 (&src)[MH], int mt0, const float (&p)[4], float f) {
#pragma unroll
        for (int mt = 0; mt < MH; ++mt) {
            attn_f4 o = O[mt0 + mt] * f;
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(src[mt].x, p[0], o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(src[mt].y, p[1], o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(src[mt].z, p[2], o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(src[mt].w, p[3], o, 0, 0, 0);
            O[mt0 + mt] = o;
        }
    };

    int it = 0;
    int tile = tile_of(0);
    if (tile < ntiles) load_k(ka, tile, 0);
    while (tile < ntiles) {
        load_k(kb, tile, KH);
        load_v(va, tile, 0);
        attn_f4 S = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KH; ++kk)
            S = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[kk], Qs[(4 * kk + quad) * ATT_BQ + col], S, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_v(vb, tile, MH);
#pragma unroll
        for (int kk = 0; kk < KH; ++kk)
            S = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[kk], Qs[(4 * (KH + kk) + quad) * ATT_BQ + col], S, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int next = tile_of(it + 1);
        if (next < ntiles) load_k(ka, next, 0);  // lands during the softmax and the PV products
        const int key0 = tile * 16 + 4 * quad;
        float sv[4] = {S.x, S.y, S.z, S.w};
        float mx = -3.0e38f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (key0 + i >= T) sv[i] = -3.0e38f;
            mx = fmaxf(mx, sv[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mnew = fmaxf(m, mx);
        const float f = __expf(m - mnew);
        float p[4], ls = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p[i] = (sv[i] <= -1.0e38f) ? 0.0f : __expf(sv[i] - mnew);
            ls += p[i];
        }
        ls += __shfl_xor(ls, 16);
        ls += __shfl_xor(ls, 32);
        l = l * f + ls;
        m = mnew;
        pv(va, 0, p, f);
        pv(vb, MH, p, f);
        __builtin_amdgcn_sched_barrier(0);
        ++it;
        tile = next;
    }

    // merge the eight per-wave states: M = max m_w, weights exp(m_w - M), fixed summation order over w
    __syncthreads();  // every wave is done with Qs
    float* wm = sm;                          // [W][16]
    float* wl = wm + ATT_WAVES * ATT_BQ;     // [W][16]
    float* wo = wl + ATT_WAVES * ATT_BQ;     // [W][C][16]
    if (quad == 0) wm[wave * ATT_BQ + col] = m;
    __syncthreads();
    float M = -3.0e38f;
#pragma unroll
    for (int w = 0; w < ATT_WAVES; ++w) M = fmaxf(M, wm[w * ATT_BQ + col]);
    const float fw = __expf(m - M);
    if (quad == 0) wl[wave * ATT_BQ + col] = l * fw;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float* dst = wo + ((size_t)wave * C + 16 * mt + 4 * quad) * ATT_BQ + col;
        dst[0 * ATT_BQ] = O[mt].x * fw;
        dst[1 * ATT_BQ] = O[mt].y * fw;
        dst[2 * ATT_BQ] = O[mt].z * fw;
        dst[3 * ATT_BQ] = O[mt].w * fw;
    }
    __syncthreads();
    for (int i = tid; i < C * ATT_BQ; i += ATT_WAVES * 64) {
        const int c = i >> 4, qi = i & 15;
        float acc = 0.0f, lt = 0.0f;
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) {
            acc += wo[((size_t)w * C + c) * ATT_BQ + qi];
            lt += wl[w * ATT_BQ + qi];
        }
        if (t0 + qi < T) out[(size_t)c * T + t0 + qi] = acc / lt;
    }
}

// ---------------------------------------------------------------- argmax + one-hot + concat
__global__ void combine_kernel(const float* __restrict__ logits, int ncls, const float* __restrict__ cont, long spatial,
                               float* __restrict__ combined, int* __restrict__ argmax_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= spatial) return;
    int best = 0;
    float bv = logits[i];
    for (int c = 1; c < ncls; ++c) {
        const float v = logits[(size_t)c * spatial + i];
        if (v > bv) { bv = v; best = c; }  // first maximum wins, as torch.argmax
    }
    for (int c = 0; c < 3; ++c) combined[(size_t)c * spatial + i] = cont[(size_t)c * spatial + i];
    for (int c = 0; c < ncls; ++c) combined[(size_t)(3 + c) * spatial + i] = (c == best) ? 1.0f : 0.0f;
    if (argmax_out) argmax_out[i] = best;
}

// save_predictions (inference_combined.py:186-195) from class ids: combined[3 + k] = (seg_pred == k)
__global__ void combine_ids_kernel(const int* __restrict__ seg, int ncls, const float* __restrict__ cont, long spatial,
                                   float* __restrict__ combined) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= spatial) return;
    const int id = seg[i];
    for (int c = 0; c < 3; ++c) combined[(size_t)c * spatial + i] = cont[(size_t)c * spatial + i];
    for (int c = 0; c < ncls; ++c) combined[(size_t)(3 + c) * spatial + i] = (c == id) ? 1.0f : 0.0f;
}

// The field exchange's wire buffer (pixie_amd/distributed.py): [n * 3 * V float32 | n * V uint8 class ids | pad to 16 B], from
// the networks' outputs in ONE pass: thread t moves float4 number t of the continuous block (3 V / 4 per scene ... the block is
// contiguous over the n scenes) and packs class ids 4 t .. 4 t + 3 into one 32-bit store.  HBM-bound: 16 B read, 13 B written per voxel.
__global__ __launch_bounds__(256) void pack_fields_kernel(const float* __restrict__ cont, const int* __restrict__ seg, long n_float, long n_vox,
                                                          unsigned char* __restrict__ wire, long wire_bytes) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float4* wf = reinterpret_cast<float4*>(wire);
    if (4 * t + 3 < n_float) wf[t] = reinterpret_cast<const float4*>(cont)[t];
    else for (long i = 4 * t; i < n_float; ++i) reinterpret_cast<float*>(wire)[i] = cont[i];
    unsigned char* wb = wire + 4 * n_float;      // (n_float = 3 n V: a multiple of 4 bytes, so 32-bit stores stay aligned when V is)
    if (4 * t + 3 < n_vox && (n_vox & 3) == 0) {
        const int4 q = reinterpret_cast<const int4*>(seg)[t];
        reinterpret_cast<unsigned*>(wb)[t] = (unsigned)(q.x & 0xff) | ((unsigned)(q.y & 0xff) << 8) | ((unsigned)(q.z & 0xff) << 16) | ((unsigned)(q.w & 0xff) << 24);
    } else {
        for (long i = 4 * t; i < n_vox && i < 4 * t + 4; ++i) wb[i] = (unsigned char)seg[i];
    }
    if (t == 0) for (long i = 4 * n_float + n_vox; i < wire_bytes; ++i) wire[i] = 0;     // the <= 15 pad bytes
}

template <int C>
static int launch_attention(const float* qkv, float* out, int T, hipStream_t st) {
    const size_t merge = ((size_t)ATT_WAVES * C * ATT_BQ + 2 * ATT_WAVES * ATT_BQ) * sizeof(float);
    const size_t qs = (size_t)C * ATT_BQ * sizeof(float);
    const size_t lds = merge > qs ? merge : qs;
    auto kern = attention_kernel<C>;
    // the attribute belongs to the (function, device) pair and a process may drive several devices: once per device
    // (not per launch: the call is not a stream operation and must stay out of HIP-graph captures)
    PX_CHECK_HIP(allow_max_dynamic_lds(reinterpret_cast<const void*>(kern)));
    hipLaunchKernelGGL(kern, dim3((T + ATT_BQ - 1) / ATT_BQ), dim3(ATT_WAVES * 64), lds, st, qkv, out, T);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- voxel-grid loader
// The reference's dataset item (WG/data_utils/my_data.py:160-224): features are stored (D,H,W,C) float16
// (pixie/voxel/voxelize.py:86,111), loaded with .astype(float32) and permuted to (C,D,H,W).  Here: one pass over the
// grid, 64 voxels x 64 channels per workgroup through an LDS tile -- 128-byte reads along C, 256-byte writes along W.
__global__ __launch_bounds__(256) void voxel_grid_to_ncdhw_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, long nvox, int C) {
    __shared__ float tile[64][65];
    const long v0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {        // r: voxel inside the tile, tx: channel
        const long v = v0 + r;
        const int c = c0 + tx;
        tile[r][tx] = (v < nvox && c < C) ? (float)src[v * C + c] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {        // r: channel inside the tile, tx: voxel
        const long v = v0 + tx;
        const int c = c0 + r;
        if (v < nvox && c < C) dst[(size_t)c * nvox + v] = tile[tx][r];
    }
}

}  // namespace pixie

using namespace pixie;

extern "C" int pixie_voxel_grid_to_ncdhw(const void* d_feat_dhwc_f16, int d, int h, int w, int channels, float* d_out_cdhw, void* stream) {
    PX_REQUIRE(d_feat_dhwc_f16 && d_out_cdhw && d > 0 && h > 0 && w > 0 && channels > 0, "pixie_voxel_grid_to_ncdhw: bad arguments");
    const long nvox = (long)d * h * w;
    hipLaunchKernelGGL(voxel_grid_to_ncdhw_kernel, dim3((unsigned)((nvox + 63) / 64), (unsigned)((channels + 63) / 64)), dim3(256), 0,
                       as_stream(stream), static_cast<const _Float16*>(d_feat_dhwc_f16), d_out_cdhw, nvox, channels);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

namespace pixie {
// the launch of pixie_channel_stats for a d_sums the caller has already zeroed (csrc/unet_exec.hip zeroes all of a forward
// pass's statistics buffers with one memset)
int channel_stats_prezeroed(const float* d_x, int channels, int64_t spatial, double* d_sums, uint32_t* d_amax, hipStream_t st) {
    // segments of >= 16 Ki elements, at most ~2048 blocks in total
    long splits = (spatial + 16383) / 16384;
    const long max_splits = std::max(1L, 2048L / channels);
    if (splits > max_splits) splits = max_splits;
    long seg = (spatial + splits - 1) / splits;
    seg = (seg + 3) & ~3L;  // keep float4 alignment of segment starts
    splits = (spatial + seg - 1) / seg;
    hipLaunchKernelGGL(channel_sums_kernel, dim3((unsigned)splits, (unsigned)channels), dim3(256), 0, st, d_x, (long)spatial, seg, d_sums,
                       reinterpret_cast<unsigned*>(d_amax));
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}
}  // namespace pixie

namespace pixie {
__global__ void zero_doubles_kernel(double* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.0;
}
}  // namespace pixie

extern "C" int pixie_channel_stats(const float* d_x, int channels, int64_t spatial, double* d_sums, uint32_t* d_amax, void* stream) {
    PX_REQUIRE(d_x && d_sums && channels > 0 && spatial > 0, "pixie_channel_stats: bad arguments");
    hipStream_t st = as_stream(stream);
    // cleared by a kernel: this call is captured into HIP graphs (pixie_amd/unet.py), and a hipMemsetAsync node did not reliably
    // precede the kernels behind it on ROCm 7.2 (unet_exec.hip, profiles/r3i_graph_replay_bisect.txt)
    hipLaunchKernelGGL(pixie::zero_doubles_kernel, dim3((unsigned)((2 * channels + 255) / 256)), dim3(256), 0, st, d_sums, 2 * channels);
    PX_CHECK_HIP(hipGetLastError());
    return pixie::channel_stats_prezeroed(d_x, channels, spatial, d_sums, d_amax, st);
}

extern "C" int pixie_channel_sums(const float* d_x, int channels, int64_t spatial, double* d_sums, void* stream) {
    return pixie_channel_stats(d_x, channels, spatial, d_sums, nullptr, stream);
}

extern "C" int pixie_norm_finalize(const double* d_sums, int channels, int64_t spatial, int mode, int groups, double eps,
                                   const float* d_weight, const float* d_bias, float* d_a, float* d_b, void* stream) {
    PX_REQUIRE(d_sums && d_a && d_b && channels > 0 && spatial > 0, "pixie_norm_finalize: bad arguments");
    PX_REQUIRE(mode == 0 || (mode == 1 && groups > 0 && channels % groups == 0), "pixie_norm_finalize: bad mode/groups");
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(cdiv(channels, 64)), dim3(64), 0, as_stream(stream), d_sums, channels,
                       static_cast<const double*>(nullptr), channels, (double)spatial, mode, groups, eps, d_weight, d_bias, d_a, d_b);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

namespace pixie {
// pixie_norm_finalize over the channel concatenation of two tensors' statistics (no copy of either)
int norm_finalize_cat(const double* d_sums0, int c0, const double* d_sums1, int c1, int64_t spatial, int mode, int groups, double eps,
                      const float* d_weight, const float* d_bias, float* d_a, float* d_b, hipStream_t st) {
    const int channels = c0 + c1;
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(cdiv(channels, 64)), dim3(64), 0, st, d_sums0, c0, d_sums1, channels, (double)spatial, mode, groups,
                       eps, d_weight, d_bias, d_a, d_b);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}
}  // namespace pixie

extern "C" int pixie_channel_affine(const float* d_x, const float* d_a, const float* d_b, float* d_y, int channels, int64_t spatial,
                                    void* stream) {
    PX_REQUIRE(d_x && d_a && d_b && d_y, "pixie_channel_affine: null argument");
    const long total = (long)channels * spatial;
    hipLaunchKernelGGL(channel_affine_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), d_x, d_a, d_b, d_y, (long)spatial, total);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int pixie_attention_forward(const float* d_qkv, float* d_out, int channels, int tokens, void* stream) {
    PX_REQUIRE(d_qkv && d_out && tokens > 0, "pixie_attention_forward: bad arguments");
    hipStream_t st = as_stream(stream);
    switch (channels) {
        case 32: return launch_attention<32>(d_qkv, d_out, tokens, st);
        case 64: return launch_attention<64>(d_qkv, d_out, tokens, st);
        case 128: return launch_attention<128>(d_qkv, d_out, tokens, st);
        case 256: return launch_attention<256>(d_qkv, d_out, tokens, st);
        default: return set_error("pixie_attention_forward: unsupported channel count %d (32/64/128/256)", channels);
    }
}

extern "C" int pixie_combine_class_ids(const int32_t* d_seg_pred, int num_classes, const float* d_cont, int64_t spatial, float* d_combined,
                                       void* stream) {
    PX_REQUIRE(d_seg_pred && d_cont && d_combined && num_classes > 0 && spatial > 0, "pixie_combine_class_ids: bad arguments");
    hipLaunchKernelGGL(combine_ids_kernel, dim3(cdiv(spatial, 256)), dim3(256), 0, as_stream(stream), d_seg_pred, num_classes, d_cont,
                       (long)spatial, d_combined);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int pixie_pack_fields(const float* d_cont, const int32_t* d_seg_pred, int64_t n_scenes, int64_t spatial, void* d_wire, int64_t wire_bytes,
                                 void* stream) {
    const int64_t n_float = 3 * n_scenes * spatial, n_vox = n_scenes * spatial;
    PX_REQUIRE(d_cont && d_seg_pred && d_wire && n_scenes > 0 && spatial > 0, "pixie_pack_fields: bad arguments");
    PX_REQUIRE((((uintptr_t)d_cont | (uintptr_t)d_seg_pred | (uintptr_t)d_wire) & 15) == 0, "pixie_pack_fields: pointers must be 16-byte aligned");
    PX_REQUIRE(wire_bytes >= 4 * n_float + n_vox && wire_bytes < 4 * n_float + n_vox + 16, "pixie_pack_fields: wire buffer of %lld bytes for %lld scenes x %lld voxels",
               (long long)wire_bytes, (long long)n_scenes, (long long)spatial);
    const int64_t threads = (std::max(n_float, n_vox) + 3) / 4;
    hipLaunchKernelGGL(pack_fields_kernel, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, as_stream(stream), d_cont, d_seg_pred, (long)n_float, (long)n_vox,
                       (unsigned char*)d_wire, (long)wire_bytes);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int pixie_combine_predictions(const float* d_logits, int num_classes, const float* d_cont, int64_t spatial, float* d_combined,
                                         int32_t* d_argmax, void* stream) {
    PX_REQUIRE(d_logits && d_cont && d_combined && num_classes > 0 && spatial > 0, "pixie_combine_predictions: bad arguments");
    hipLaunchKernelGGL(combine_kernel, dim3(cdiv(spatial, 256)), dim3(256), 0, as_stream(stream), d_logits, num_classes, d_cont, (long)spatial,
                       d_combined, d_argmax);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}
