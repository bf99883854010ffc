// Micro-benchmark: MFMA 32x32x16 f16 issue rate of a 4-wave workgroup running the conv kernel's tap loop
// (24 MFMAs on 8 accumulators per tap) with the B fragments (a) kept in registers, (b) re-read from LDS every tap
// (8 x ds_read_b128), (c) re-read from LDS one tap ahead (software prefetch), at 1 or 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <string>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned rnd_half2(unsigned x) {   // two fp16 values in [-2, 2) with random mantissas
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (x & 0x83ff83ffu) | 0x3c003c00u;
}
// random = 2 / 3: the `lo` operands keep only their top 3 / 0 mantissa bits (how much of the power limit is the lo terms?)
__device__ __host__ inline unsigned lomask(bool is_lo, int random) { return (!is_lo || random < 2) ? 0xffffffffu : (random == 2 ? 0xff80ff80u : 0xfc00fc00u); }
template <int MODE, int OCC>
__global__ __launch_bounds__(256, OCC) void k(const uint4* __restrict__ w, float* __restrict__ out, int taps, int lds_units, int random) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < lds_units; i += 256)
        lds[i] = random ? make_uint4(rnd_half2(4 * i) & lomask(i >= 600, random), rnd_half2(4 * i + 1) & lomask(i >= 600, random), rnd_half2(4 * i + 2) & lomask(i >= 600, random), rnd_half2(4 * i + 3) & lomask(i >= 600, random))
                        : make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    __syncthreads();
    f32x16 acc[2][4];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    f16x8 ah[2], al[2], bh[4], bl[4];
    for (int m = 0; m < 2; ++m) { ah[m] = __builtin_bit_cast(f16x8, w[lane + m * 64]); al[m] = __builtin_bit_cast(f16x8, w[lane + 128 + m * 64]); }
    int off = wave * 128 + lane;
    for (int n = 0; n < 4; ++n) { bh[n] = __builtin_bit_cast(f16x8, lds[off + n * 32]); bl[n] = __builtin_bit_cast(f16x8, lds[off + n * 32 + 600]); }
#pragma unroll 1
    for (int t = 0; t < taps; ++t) {
        f16x8 bhn[4], bln[4], ahn[2], aln[2];
        const int o2 = off + (t % 27) + 1;
        if (MODE >= 3) {   // the kernel's A-fragment stream: fragments of the next tap from global memory (L1/L2-hot)
            const uint4* wn = w + (size_t)(((t + 1) % 27) * 256);
            for (int m = 0; m < 2; ++m) { ahn[m] = __builtin_bit_cast(f16x8, wn[lane + m * 64]); aln[m] = __builtin_bit_cast(f16x8, wn[lane + 128 + m * 64]); }
            if (MODE >= 4 && (t % 27) == 0) { __syncthreads(); __syncthreads(); }
        }
        if (MODE == 1 || MODE >= 3) { for (int n = 0; n < 4; ++n) { bh[n] = __builtin_bit_cast(f16x8, lds[o2 + n * 32]); bl[n] = __builtin_bit_cast(f16x8, lds[o2 + n * 32 + 600]); } }
        if (MODE == 2) { for (int n = 0; n < 4; ++n) { bhn[n] = __builtin_bit_cast(f16x8, lds[o2 + n * 32]); bln[n] = __builtin_bit_cast(f16x8, lds[o2 + n * 32 + 600]); } }
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], acc[m][n], 0, 0, 0);
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], acc[m][n], 0, 0, 0);
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
        if (MODE == 2) { for (int n = 0; n < 4; ++n) { bh[n] = bhn[n]; bl[n] = bln[n]; } }
        if (MODE >= 3) { for (int m = 0; m < 2; ++m) { ah[m] = ahn[m]; al[m] = aln[m]; } }
    }
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int OCC>
void run(const char* name, const uint4* w, float* out, int wgs, size_t lds_bytes, int random = 0) {
    const int taps = 27 * 64;
    auto kern = k<MODE, OCC>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds_bytes, 0, w, out, taps, 1300, random);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds_bytes, 0, w, out, taps, 1300, random);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)wgs * 4 * taps * 24 * 2.0 * 32 * 32 * 16;
    printf("%-44s %7.3f ms  %7.1f TFLOP/s f16 MFMA (%.1f %% of 2500)\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0);
}

// `mfma_lds.exe sustain <seconds>`: the random-operand loop (B from LDS + A from global, 2 WG/CU) back to back for that long,
// so that the board power can be sampled around it (scripts/conv_energy.py): prints launches, ms per launch, FLOP per launch.
static int sustain(double seconds) {
    uint4* wr; float* out;
    CK(hipMalloc(&wr, 8192 * 16)); CK(hipMalloc(&out, 4096 * 256 * 4));
    unsigned* h = (unsigned*)malloc(8192 * 16);
    unsigned x = 12345u;
    for (int i = 0; i < 8192 * 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 3) & 0x83ff83ffu) | 0x3c003c00u; }
    CK(hipMemcpy(wr, h, 8192 * 16, hipMemcpyHostToDevice));
    auto kern = k<3, 2>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int taps = 27 * 64, wgs = 2048;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 78 * 1024, 0, wr, out, taps, 1300, 1);
    CK(hipDeviceSynchronize());
    long launches = 0; float total_ms = 0.f;
    while (total_ms < seconds * 1e3) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 78 * 1024, 0, wr, out, taps, 1300, 1);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms; launches += 50;
    }
    const double flop = (double)wgs * 4 * taps * 24 * 2.0 * 32 * 32 * 16;
    printf("SUSTAIN launches %ld ms_per_launch %.4f flop_per_launch %.6e tflops %.1f\n", launches, total_ms / launches, flop, flop / (total_ms / launches) / 1e9);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 3 && std::string(argv[1]) == "sustain") return sustain(atof(argv[2]));
    uint4* w; float* out;
    CK(hipMalloc(&w, 8192 * 16)); CK(hipMemset(w, 0x3c, 8192 * 16)); CK(hipMalloc(&out, 4096 * 256 * 4));
    run<0, 2>("B in registers, 2 WG/CU (78 KB LDS each)", w, out, 2048, 78 * 1024);
    run<1, 2>("B from LDS each tap, 2 WG/CU", w, out, 2048, 78 * 1024);
    run<2, 2>("B from LDS one tap ahead, 2 WG/CU", w, out, 2048, 78 * 1024);
    run<0, 1>("B in registers, 1 WG/CU (156 KB LDS)", w, out, 1024, 156 * 1024);
    run<1, 1>("B from LDS each tap, 1 WG/CU", w, out, 1024, 156 * 1024);
    run<2, 1>("B from LDS one tap ahead, 1 WG/CU", w, out, 1024, 156 * 1024);
    run<3, 2>("B from LDS + A from global each tap, 2 WG/CU", w, out, 2048, 78 * 1024);
    {   // the same loops on random operands (the constant patterns above toggle almost no bits in the multipliers)
        uint4* wr; CK(hipMalloc(&wr, 8192 * 16));
        unsigned* h = (unsigned*)malloc(8192 * 16);
        unsigned x = 12345u;
        for (int i = 0; i < 8192 * 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 3) & 0x83ff83ffu) | 0x3c003c00u; }
        CK(hipMemcpy(wr, h, 8192 * 16, hipMemcpyHostToDevice));
        run<0, 2>("RANDOM data: B in registers, 2 WG/CU", wr, out, 2048, 78 * 1024, 1);
        run<3, 2>("RANDOM data: B from LDS + A from global, 2 WG/CU", wr, out, 2048, 78 * 1024, 1);
        run<3, 1>("RANDOM data: B from LDS + A from global, 1 WG/CU", wr, out, 1024, 156 * 1024, 1);
        run<0, 1>("RANDOM data: B in registers, 1 WG/CU", wr, out, 1024, 156 * 1024, 1);
        for (int mode = 2; mode <= 3; ++mode) {
            for (int i = 0; i < 8192 * 4; ++i) if (((i / 4) % 256) >= 128) h[i] &= lomask(true, mode);
            CK(hipMemcpy(wr, h, 8192 * 16, hipMemcpyHostToDevice));
            run<3, 2>(mode == 2 ? "RANDOM hi, lo operands 3 mantissa bits, 2 WG/CU" : "RANDOM hi, lo operands 0 mantissa bits, 2 WG/CU", wr, out, 2048, 78 * 1024, mode);
        }
    }
    run<4, 2>("  ... + 2 barriers per 27 taps, 2 WG/CU", w, out, 2048, 78 * 1024);
    run<0, 2>("B in registers, 4 WG/CU (32 KB LDS each)", w, out, 4096, 32 * 1024);
    run<1, 2>("B from LDS each tap, 4 WG/CU (32 KB)", w, out, 4096, 32 * 1024);
    return 0;
}
