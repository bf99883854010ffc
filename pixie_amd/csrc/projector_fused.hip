// pixie_amd/csrc/projector_fused.hip -- the first projector convolution of BOTH networks straight from the voxel grid.
//
// The reference stores a scene's features as a (D, H, W, C) float16 array (pixie/voxel/voxelize.py:86,111; C = 768 CLIP
// channels in the shipped configuration, config/training/default.yaml:5,29), converts it to float32, permutes it to
// (C, D, H, W) (WG/data_utils/my_data.py:160-224) and then runs FeatureProjector.net[0] -- Conv3d(C, 128, 1) -- once in
// the segmentation network and once in the regression network (WG/models/module/diffusion_network.py:556-560).  That is
// one 2-byte read, one 4-byte write and two 4-byte reads of every feature: 14 B per feature, 2.8 GB per 64^3 x 768 scene.
//
// Here the grid is consumed where it lies.  A 1x1x1 convolution over a channels-last grid is a plain GEMM
//     Out[net][co][v] = sum_c W[net][co][c] * X[v][c],      M = c_out of all networks, N = voxels, K = C,
// whose B operand (16 voxels x 8 consecutive channels per lane = one 16-byte load) is exactly how the file is laid out,
// and whose inputs are exact fp16 numbers: only the weights need the hi/lo split of conv3d_f16x3.hip, so two f16 MFMAs
// per product reproduce the fp32 result to 2^-22.  Both networks' weights are the A operand of ONE launch: the grid is
// read once (2 B per feature) and two (128, D, H, W) float32 tensors are written.
//
// Workgroup = 4 waves over a tile of (all output channels) x (64 or 128 voxels); per 64-channel chunk the voxel tile is
// staged in LDS as [8-channel group][voxel] x 16 B (coalesced 128-byte row segments in, conflict-free ds_read_b128 B
// fragments out); A fragments stream from L2 (the weights of both networks are 768 KB at C = 768).
#include <hip/hip_runtime.h>

#include "../../include/pixie_hip.h"
#include "common.h"

namespace pixie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kPW16HeaderU4 = 4;   // header of pixie_conv_pack_weights_f16x2: [0].x = bits of 1 / s_w
constexpr int kPMaxNets = 2;
constexpr int kPKC = 64;           // channels per LDS chunk (4 MFMA K steps)

struct ProjArgs {
    const uint4* feat;             // (voxels, C) fp16, 8 channels per uint4
    long voxels;
    int C, KG;                     // channels, C / 8
    int nets, coutp, cout;         // every network: cout real output channels, padded to coutp (multiple of 32)
    const uint4* w16[kPMaxNets];   // packed by pixie_conv_pack_weights_f16x2 (ksize 1)
    const float* bias[kPMaxNets];
    float* out[kPMaxNets];         // (cout, voxels) float32
    int waves_m, nv;               // waves along M (each 64 rows), voxels per workgroup = (4 / waves_m) * 64
};

__global__ __launch_bounds__(256, 2) void projector_conv0_kernel(ProjArgs A) {
    extern __shared__ uint4 xs[];                      // [8][nv + 1] x 16 B  (the +1 staggers the banks of the 8 groups)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5, l31 = lane & 31;
    const int wm = wave % A.waves_m, wn = wave / A.waves_m;
    const int row0 = wm * 64;                          // first output row of this wave among nets * coutp rows
    const int net = row0 / A.coutp, co0 = row0 % A.coutp;
    const long v0 = (long)blockIdx.x * A.nv;
    const int vw = wn * 64;                            // this wave's first voxel inside the tile
    const int stride = A.nv + 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    const uint4* wHi = A.w16[net] + kPW16HeaderU4 + co0 + l31;
    const size_t lo_off = (size_t)A.KG * A.coutp;      // the lo plane follows the hi plane (one tap)
    const int units = A.nv * 8;                        // 16-byte units of one chunk of the tile

    for (int c_base = 0; c_base < A.C; c_base += kPKC) {
        __syncthreads();                               // previous chunk consumed
        for (int u = tid; u < units; u += 256) {
            const int vox = u >> 3, kg = u & 7;
            const long v = v0 + vox;
            uint4 q = make_uint4(0u, 0u, 0u, 0u);
            if (v < A.voxels && c_base + 8 * kg < A.C) q = A.feat[v * A.KG + (c_base >> 3) + kg];
            xs[kg * stride + vox] = q;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < kPKC / 16; ++ks) {
            const int kg = (c_base >> 3) + 2 * ks + kh;          // this lane's 8-channel group
            if (c_base + 16 * ks >= A.C) break;                  // (uniform) partial last chunk
            f16x8 ah[2], al[2], b[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const uint4 h = wHi[(size_t)kg * A.coutp + mb * 32];
                const uint4 l = wHi[lo_off + (size_t)kg * A.coutp + mb * 32];
                ah[mb] = __builtin_bit_cast(f16x8, h);
                al[mb] = __builtin_bit_cast(f16x8, l);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) b[nb] = __builtin_bit_cast(f16x8, xs[(2 * ks + kh) * stride + vw + nb * 32 + l31]);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mb], b[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], b[nb], acc[mb][nb], 0, 0, 0);
        }
    }

    // C/D layout of v_mfma_f32_32x32x16_f16: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const float inv = __uint_as_float(A.w16[net][0].x);
    float* out = A.out[net];
    const float* bias = A.bias[net];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (co >= A.cout) continue;
            const float bv = bias ? bias[co] : 0.0f;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const long v = v0 + vw + nb * 32 + l31;
                if (v < A.voxels) out[(size_t)co * A.voxels + v] = acc[mb][nb][r] * inv + bv;   // 128-byte runs per row
            }
        }
}

}  // namespace pixie

using namespace pixie;

extern "C" int pixie_projector_conv0(const void* d_feat_dhwc_f16, int64_t voxels, int channels, int n_networks, const void* const* d_w16,
                                     const float* const* d_bias, float* const* d_out, int c_out, void* stream) {
    PX_REQUIRE(d_feat_dhwc_f16 && d_w16 && d_out && voxels > 0, "pixie_projector_conv0: null argument");
    PX_REQUIRE(n_networks >= 1 && n_networks <= kPMaxNets, "pixie_projector_conv0: 1 or 2 networks");
    PX_REQUIRE(channels > 0 && channels % 16 == 0, "pixie_projector_conv0: the channel count must be a multiple of 16 (got %d)", channels);
    const int coutp = pixie_conv_cout_padded(c_out);
    PX_REQUIRE(c_out > 0 && coutp % 64 == 0, "pixie_projector_conv0: c_out must pad to a multiple of 64 (got %d -> %d)", c_out, coutp);
    const int rows = n_networks * coutp;
    PX_REQUIRE(rows == 64 || rows == 128 || rows == 256, "pixie_projector_conv0: %d x %d output rows are not a tiling this kernel has", n_networks, coutp);
    ProjArgs a{};
    a.feat = static_cast<const uint4*>(d_feat_dhwc_f16);
    a.voxels = voxels; a.C = channels; a.KG = channels / 8;
    a.nets = n_networks; a.coutp = coutp; a.cout = c_out;
    for (int n = 0; n < n_networks; ++n) {
        PX_REQUIRE(d_w16[n] && d_out[n], "pixie_projector_conv0: null weights / output for network %d", n);
        a.w16[n] = static_cast<const uint4*>(d_w16[n]);
        a.bias[n] = d_bias ? d_bias[n] : nullptr;
        a.out[n] = d_out[n];
    }
    a.waves_m = rows / 64;
    a.nv = (4 / a.waves_m) * 64;
    const size_t lds = (size_t)8 * (a.nv + 1) * sizeof(uint4);
    const long tiles = (voxels + a.nv - 1) / a.nv;
    hipLaunchKernelGGL(projector_conv0_kernel, dim3((unsigned)tiles), dim3(256), lds, as_stream(stream), a);
    PX_CHECK_HIP(hipGetLastError());
    return 0;
}
