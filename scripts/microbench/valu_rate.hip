// Micro-benchmark: VALU issue cost per wave64 instruction on gfx950, as a function of the instruction kind and of the
// number of waves resident per SIMD.  Answers "is the MPM block kernel (≈1800 VALU instructions per wave, 3 waves per
// SIMD) VALU-issue-bound?": cycles per instruction = kernel cycles / (instructions per wave x waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/valu_rate.hip -o scripts/microbench/valu_rate.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kIters = 2000;
constexpr int kChains = 8;    // independent dependency chains per lane: 8 x 4-cycle latency is hidden even for one wave

// MODE 0: v_fma_f32   1: v_pk_fma_f32 (2 FMAs per lane)   2: v_mul_f32 + v_add_f32 pairs   3: v_cndmask_b32
// MODE 4: v_rcp_f32 (transcendental)   5: v_add_f64   6: v_cvt_f64_f32 + v_add_f64   7: v_mov_b32 (via readfirstlane-free copy)
template <int MODE>
__global__ __launch_bounds__(256) void valu_kernel(float* __restrict__ out, float a, float b, int n_iter) {
    float x[kChains];
    float2 p[kChains];
    double d[kChains];
#pragma unroll
    for (int c = 0; c < kChains; ++c) { x[c] = a + c + threadIdx.x * 1e-3f; p[c] = make_float2(x[c], x[c] + 1.f); d[c] = x[c]; }
    for (int it = 0; it < n_iter; ++it) {
#pragma unroll
        for (int c = 0; c < kChains; ++c) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(make_float2(a, a)), "v"(make_float2(b, b)));
            if (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(x[c]) : "v"(a), "v"(b));
            if (MODE == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(a) : );
            if (MODE == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[c]));
            if (MODE == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"((double)b));
            if (MODE == 6) asm volatile("v_cvt_f64_f32 %0, %1\n\tv_add_f64 %0, %0, %2" : "=&v"(d[c]) : "v"(x[c]), "v"((double)b));
            if (MODE == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(x[c]) : "v"(p[c].x));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < kChains; ++c) s += x[c] + p[c].x + p[c].y + (float)d[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int insts_per_slot, float* d_out, double ghz) {
    // one workgroup = 4 waves = one wave per SIMD of a CU; blocks_per_cu workgroups per CU -> that many waves per SIMD
    for (int per_cu : {1, 2, 3, 4, 8}) {
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
        printf("%-28s waves/SIMD %d: %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.1f GHz nominal)\n", name, per_cu, ms,
               cycles / (insts_per_wave * per_cu), ghz);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    const double ghz = 2.4;
    run<0>("v_fma_f32", 1, d_out, ghz);
    run<1>("v_pk_fma_f32", 1, d_out, ghz);
    run<2>("v_mul_f32 + v_add_f32", 2, d_out, ghz);
    run<3>("v_cndmask_b32", 1, d_out, ghz);
    run<4>("v_rcp_f32", 1, d_out, ghz);
    run<5>("v_add_f64", 1, d_out, ghz);
    run<6>("v_cvt_f64_f32 + v_add_f64", 2, d_out, ghz);
    run<7>("v_mov_b32", 1, d_out, ghz);
    hipFree(d_out);
    return 0;
}
