// Micro-benchmark: candidate implementations of the P2G tile scatter of the fused MPM kernel (27 stencil nodes x
// (momentum.xyz, mass) per particle into an 8^3-node LDS tile), isolated from the rest of the kernel.  Lanes of a wave sit
// in distinct cells (the production ordering), 256-thread workgroups, several workgroups per CU.
//   V0  4 x ds_add_u64 per node on a workgroup-shared tile, values through the double-precision fixed-point trick (today)
//   V1  per-wave private float4 tile, plain read-modify-write: ds_read_b128 + 4 v_add + ds_write_b128 (one pass)
//   V2  V1 in two passes of half the lanes each (what duplicate cells inside a wave would cost)
//   V3  V0 with every second lane switched off (pairs pre-merged in registers)
//   V4  per-wave private tile, 2 x (ds_read_b64 + 2 v_add + ds_write_b64)
// Build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/microbench/scatter_variants.hip -o scripts/microbench/scatter_variants.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int TN = 512;
constexpr double kMagic = 6755399441055744.0;

__device__ __forceinline__ unsigned long long to_fixed(float x) { return (unsigned long long)__double_as_longlong((double)x + kMagic); }

template <int V>
__global__ __launch_bounds__(256) void scatter_kernel(const int* __restrict__ base_idx, float* __restrict__ out, int rounds) {
    extern __shared__ unsigned char smem[];
    unsigned long long* ta = reinterpret_cast<unsigned long long*>(smem);          // V0/V3: [4][TN] u64 = 16 KB
    float4* tw = reinterpret_cast<float4*>(smem) + (threadIdx.x >> 6) * TN;        // V1/V2/V4: [4 waves][TN] float4 = 32 KB
    const int tid = threadIdx.x;
    if (V == 0 || V == 3) { for (int i = tid; i < 4 * TN; i += 256) ta[i] = 0ull; }
    else { for (int i = tid; i < 4 * TN; i += 256) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
    __syncthreads();
    const int b0 = base_idx[blockIdx.x * 256 + tid];
    float val = 1.0f + tid * 1e-3f;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int idx = b0 + (i * 8 + j) * 8 + k;
                    const float mx = val * (float)(1 + i), my = val * (float)(2 + j), mz = val * (float)(3 + k), mm = val * 0.5f;
                    if (V == 0 || (V == 3 && (tid & 1) == 0)) {
                        atomicAdd(&ta[0 * TN + idx], to_fixed(mx));
                        atomicAdd(&ta[1 * TN + idx], to_fixed(my));
                        atomicAdd(&ta[2 * TN + idx], to_fixed(mz));
                        atomicAdd(&ta[3 * TN + idx], to_fixed(mm));
                    }
                    if (V == 1) {
                        float4 q = tw[idx];
                        q.x += mx; q.y += my; q.z += mz; q.w += mm;
                        tw[idx] = q;
                        __builtin_amdgcn_wave_barrier();   // lanes of the wave alias across stencil offsets: keep the order
                    }
                    if (V == 2) {
#pragma unroll
                        for (int pass = 0; pass < 2; ++pass) {
                            if (((tid >> 5) & 1) == pass) {
                                float4 q = tw[idx];
                                q.x += mx; q.y += my; q.z += mz; q.w += mm;
                                tw[idx] = q;
                            }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                    if (V == 4) {
                        float2* t2 = reinterpret_cast<float2*>(tw);
                        float2 a = t2[2 * idx], b = t2[2 * idx + 1];
                        a.x += mx; a.y += my; b.x += mz; b.y += mm;
                        t2[2 * idx] = a; t2[2 * idx + 1] = b;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
        val += 1e-6f;
    }
    __syncthreads();
    float s = 0.f;
    if (V == 0 || V == 3) { for (int i = tid; i < 4 * TN; i += 256) s += (float)ta[i]; }
    else { for (int i = tid; i < 4 * TN; i += 256) { const float4 q = reinterpret_cast<float4*>(smem)[i]; s += q.x + q.y + q.z + q.w; } }
    out[blockIdx.x * 256 + tid] = s + val;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V>
void run(const char* name, const int* d_base, float* d_out, int blocks, int rounds, size_t lds) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(scatter_kernel<V>, dim3(blocks), dim3(256), lds, 0, d_base, d_out, rounds);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(scatter_kernel<V>, dim3(blocks), dim3(256), lds, 0, d_base, d_out, rounds);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_particles = (double)blocks * 4 * rounds;      // one "particle-wave" = 64 particles x 27 nodes x 4 values
    printf("%-58s %8.3f ms   %7.1f cycles per particle-wave per CU (2.4 GHz, 256 CUs)   = %5.1f us per 15625 particle-waves (1 M particles)\n", name, ms,
           ms * 1e-3 * 2.4e9 * 256 / wave_particles, ms * 1e3 * 15625.0 / wave_particles);
}

int main() {
    const int blocks = 5120, rounds = 16;
    int* h = (int*)malloc(sizeof(int) * blocks * 256);
    // 64 lanes of a wave = the 64 cells of a 4^3 block, base node of cell (cx,cy,cz) = (cx,cy,cz) in the 8^3 tile
    for (int i = 0; i < blocks * 256; ++i) { const int c = i % 64; h[i] = ((c >> 4) * 8 + ((c >> 2) & 3)) * 8 + (c & 3); }
    int* d_base; float* d_out;
    CK(hipMalloc(&d_base, sizeof(int) * blocks * 256)); CK(hipMalloc(&d_out, sizeof(float) * blocks * 256));
    CK(hipMemcpy(d_base, h, sizeof(int) * blocks * 256, hipMemcpyHostToDevice));
    run<0>("V0 4 x ds_add_u64 per node, shared tile (today)", d_base, d_out, blocks, rounds, 16384);
    run<1>("V1 per-wave float4 tile, read_b128 + add + write_b128", d_base, d_out, blocks, rounds, 32768);
    run<2>("V2 as V1 in two half-wave passes", d_base, d_out, blocks, rounds, 32768);
    run<3>("V3 as V0 with every second lane off", d_base, d_out, blocks, rounds, 16384);
    run<4>("V4 per-wave tile, 2 x (read_b64 + add + write_b64)", d_base, d_out, blocks, rounds, 32768);
    // the production order is round-robin over cells but a wave may hold a few duplicates: random cells as the worst case
    srand(1);
    for (int i = 0; i < blocks * 256; ++i) { int lx = rand() % 4, ly = rand() % 4, lz = rand() % 4; h[i] = (lx * 8 + ly) * 8 + lz; }
    CK(hipMemcpy(d_base, h, sizeof(int) * blocks * 256, hipMemcpyHostToDevice));
    run<0>("V0 random cells (conflicts)", d_base, d_out, blocks, rounds, 16384);
    return 0;
}
