// scripts/microbench/valu_rate2.hip -- issue cost of the "other" VALU instructions the fused MPM kernel is made of: selects,
// compares, min/max, the IEEE-division helper instructions, sqrt/rsq, conversions.  Same method as valu_rate.hip: 8 independent
// chains per lane, 1..8 waves per SIMD, nominal cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int kIters = 2000;
constexpr int kChains = 8;

template <int MODE>
__global__ __launch_bounds__(256) void valu_kernel(float* __restrict__ out, float a, float b, int n_iter) {
    float x[kChains];
#pragma unroll
    for (int c = 0; c < kChains; ++c) x[c] = a + c + threadIdx.x * 1e-3f;
    unsigned long long m = 0x5555555555555555ull ^ (unsigned long long)blockIdx.x;
    asm volatile("s_mov_b64 vcc, %0" :: "s"(m) : "vcc");
    for (int it = 0; it < n_iter; ++it) {
#pragma unroll
        for (int c = 0; c < kChains; ++c) {
            if (MODE == 0) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(a));
            if (MODE == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "s"(m));
            if (MODE == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
            if (MODE == 3) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x[c]), "v"(a) : "vcc");
            if (MODE == 4) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(a) : "vcc");
            if (MODE == 5) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x[c]) : "v"(a) : "vcc");
            if (MODE == 6) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
            if (MODE == 7) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
            if (MODE == 8) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[c]));
            if (MODE == 9) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[c]));
            if (MODE == 10) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
            if (MODE == 11) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x[c]));
            if (MODE == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
            if (MODE == 13) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
            if (MODE == 14) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
            if (MODE == 15) asm volatile("v_cmp_class_f32 vcc, %0, %1" :: "v"(x[c]), "v"(a) : "vcc");
            if (MODE == 16) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[c]) : "v"(3));
            if (MODE == 17) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x[c]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < kChains; ++c) s += x[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int insts_per_slot, float* d_out, double ghz) {
    for (int per_cu : {1, 2, 4, 8}) {
        const int blocks = 256 * per_cu;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(valu_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f, 0.5f, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(valu_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f, 0.5f, kIters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_wave = (double)kIters * kChains * insts_per_slot;
        const double cycles = ms * 1e-3 * ghz * 1e9;
        printf("%-34s waves/SIMD %d: %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.1f GHz nominal)\n", name, per_cu, ms,
               cycles / (insts_per_wave * per_cu), ghz);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    const double ghz = 2.4;
    run<0>("v_cndmask_b32 (vcc)", 1, d_out, ghz);
    run<1>("v_cndmask_b32_e64 (sgpr mask)", 1, d_out, ghz);
    run<2>("v_max_f32", 1, d_out, ghz);
    run<3>("v_cmp_gt_f32 -> vcc", 1, d_out, ghz);
    run<4>("v_cmp_gt_f32 + v_cndmask_b32", 2, d_out, ghz);
    run<5>("v_div_scale_f32", 1, d_out, ghz);
    run<6>("v_div_fmas_f32", 1, d_out, ghz);
    run<7>("v_div_fixup_f32", 1, d_out, ghz);
    run<8>("v_sqrt_f32", 1, d_out, ghz);
    run<9>("v_rsq_f32", 1, d_out, ghz);
    run<10>("v_max3_f32", 1, d_out, ghz);
    run<11>("v_cvt_f32_i32", 1, d_out, ghz);
    run<12>("v_mul_f32", 1, d_out, ghz);
    run<13>("v_fmac_f32", 1, d_out, ghz);
    run<14>("v_and_b32", 1, d_out, ghz);
    run<15>("v_cmp_class_f32 -> vcc", 1, d_out, ghz);
    run<16>("v_ldexp_f32", 1, d_out, ghz);
    run<17>("v_bfi_b32", 1, d_out, ghz);
    hipFree(d_out);
    return 0;
}
