// Micro-benchmark: throughput of LDS atomic flavours in the access pattern of the MPM P2G scatter
// (each lane adds 27 stencil nodes x 4 words into an 8^3-node SoA tile).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int TN = 512;

template <int MODE>
__global__ __launch_bounds__(256) void scatter_kernel(const int* __restrict__ base_idx, float* __restrict__ out, int rounds) {
    __shared__ float tf[4][TN];
    __shared__ unsigned long long tq[4][TN];
    double* td = reinterpret_cast<double*>(&tq[0][0]);
    unsigned* tu = reinterpret_cast<unsigned*>(&tf[0][0]);
    const int tid = threadIdx.x;
    for (int i = tid; i < TN; i += 256) { for (int c = 0; c < 4; ++c) { tf[c][i] = 0.f; tq[c][i] = 0ull; } }
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
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float v = val * (float)(c + 1 + i + j + k);
                        if (MODE == 0) atomicAdd(&tf[c][idx], v);                                   // ds_add_f32
                        if (MODE == 1) atomicAdd(&tu[c * TN + idx], (unsigned)(int)(v * 1024.f));     // ds_add_u32
                        if (MODE == 2) atomicAdd(&tq[c][idx], (unsigned long long)(long long)(v * 1048576.f));  // ds_add_u64
                        if (MODE == 3) tf[c][idx] += v;                                               // racy RMW (timing only)
                        if (MODE == 4) { float old = atomicAdd(&tf[c][idx], v); val += old * 1e-30f; }  // returning
                        if (MODE == 5) atomicAdd(&td[c * TN + idx], (double)v);                          // ds_add_f64
                    }
                }
        val += 1e-6f;
    }
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < TN; i += 256) for (int c = 0; c < 4; ++c) s += tf[c][i] + (float)tq[c][i];
    out[blockIdx.x * 256 + tid] = s + val;
}

// global fp32 atomics in the flush pattern: each workgroup adds a (mostly distinct) 216-node x 4-word tile into a grid
template <int MODE>
__global__ __launch_bounds__(256) void flush_kernel(float* __restrict__ grid, int ng, int nbk) {
    const int b = blockIdx.x % (nbk * nbk * nbk);
    const int bz = b % nbk, by = (b / nbk) % nbk, bx = b / (nbk * nbk);
    for (int idx = threadIdx.x; idx < TN; idx += 256) {
        const int lz = idx & 7, ly = (idx >> 3) & 7, lx = idx >> 6;
        if (lx >= 6 || ly >= 6 || lz >= 6) continue;
        const int gx = bx * 4 + lx, gy = by * 4 + ly, gz = bz * 4 + lz;
        if (gx >= ng || gy >= ng || gz >= ng) continue;
        float* cell = grid + 4 * (((size_t)gx * ng + gy) * ng + gz);
        if (MODE == 0) { unsafeAtomicAdd(cell + 0, 1.f); unsafeAtomicAdd(cell + 1, 2.f); unsafeAtomicAdd(cell + 2, 3.f); unsafeAtomicAdd(cell + 3, 4.f); }
        if (MODE == 1) { cell[0] += 1.f; cell[1] += 2.f; cell[2] += 3.f; cell[3] += 4.f; }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
void run_scatter(const char* name, const int* d_base, float* d_out, int blocks, int rounds) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(scatter_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_base, d_out, rounds);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(scatter_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_base, d_out, rounds);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)blocks * 256 * rounds * 108;
    printf("%-28s %8.3f ms  %8.2f G lane-atomics/s  (%.2f cycles per wave-instruction per CU at 2.4 GHz, 256 CUs)\n", name, ms,
           ops / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / (ops / 64));
}

int main() {
    const int blocks = 4096, rounds = 8;
    int* h = (int*)malloc(sizeof(int) * blocks * 256);
    srand(1);
    for (int i = 0; i < blocks * 256; ++i) { int lx = rand() % 6, ly = rand() % 6, lz = rand() % 6; h[i] = (lx * 8 + ly) * 8 + lz; }
    int* d_base; float* d_out;
    CK(hipMalloc(&d_base, sizeof(int) * blocks * 256)); CK(hipMalloc(&d_out, sizeof(float) * blocks * 256));
    CK(hipMemcpy(d_base, h, sizeof(int) * blocks * 256, hipMemcpyHostToDevice));
    run_scatter<0>("ds_add_f32 random cells", d_base, d_out, blocks, rounds);
    run_scatter<1>("ds_add_u32 random cells", d_base, d_out, blocks, rounds);
    run_scatter<2>("ds_add_u64 random cells", d_base, d_out, blocks, rounds);
    run_scatter<3>("plain RMW (racy)", d_base, d_out, blocks, rounds);
    run_scatter<4>("ds_add_rtn_f32", d_base, d_out, blocks, rounds);
    run_scatter<5>("ds_add_f64 random cells", d_base, d_out, blocks, rounds);
    // cell-sorted lanes: runs of 12 consecutive lanes share a cell (same 27 addresses)
    for (int i = 0; i < blocks * 256; ++i) { int cell = (i / 12) * 2654435761u % 216; h[i] = ((cell / 36) * 8 + (cell / 6) % 6) * 8 + cell % 6; }
    CK(hipMemcpy(d_base, h, sizeof(int) * blocks * 256, hipMemcpyHostToDevice));
    run_scatter<0>("ds_add_f32 cell-sorted", d_base, d_out, blocks, rounds);
    run_scatter<1>("ds_add_u32 cell-sorted", d_base, d_out, blocks, rounds);
    run_scatter<2>("ds_add_u64 cell-sorted", d_base, d_out, blocks, rounds);
    // all lanes distinct addresses, conflict-free banks
    for (int i = 0; i < blocks * 256; ++i) h[i] = (i % 64) % 6 + 8 * (((i % 64) / 6) % 6) + 64 * ((i % 64) / 36);
    CK(hipMemcpy(d_base, h, sizeof(int) * blocks * 256, hipMemcpyHostToDevice));
    run_scatter<0>("ds_add_f32 distinct cells", d_base, d_out, blocks, rounds);
    run_scatter<1>("ds_add_u32 distinct cells", d_base, d_out, blocks, rounds);

    // global flush
    const int ng = 120, nbk = 30;
    float* d_grid; CK(hipMalloc(&d_grid, sizeof(float) * 4 * ng * ng * ng)); CK(hipMemset(d_grid, 0, sizeof(float) * 4 * ng * ng * ng));
    for (int mode = 0; mode < 2; ++mode) {
        for (int wgs : {528, 5560, 27000}) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            if (mode == 0) hipLaunchKernelGGL(flush_kernel<0>, dim3(wgs), dim3(256), 0, 0, d_grid, ng, nbk);
            else hipLaunchKernelGGL(flush_kernel<1>, dim3(wgs), dim3(256), 0, 0, d_grid, ng, nbk);
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(flush_kernel<0>, dim3(wgs), dim3(256), 0, 0, d_grid, ng, nbk);
            else hipLaunchKernelGGL(flush_kernel<1>, dim3(wgs), dim3(256), 0, 0, d_grid, ng, nbk);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("flush %s %5d WGs x 864 words: %8.2f us  %7.2f G atomics/s\n", mode == 0 ? "atomic" : "plain ", wgs, ms * 1e3, wgs * 864.0 / ms / 1e6);
        }
    }
    return 0;
}
