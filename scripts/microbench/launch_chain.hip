// Floor of a two-kernel substep: how long does a chain of DEPENDENT (same-stream) launch pairs take when the kernels do
// (almost) nothing -- with the launch geometry of the MPM substep at 100 k particles (526 x 256 threads, then 576 x 64) and at
// 1 M (5557 x 256, 3744 x 64)?  Kernel A writes one float4 per workgroup, kernel B reads it and writes another: a real
// producer/consumer pair, so the end-of-kernel cache write-back / invalidate between them is what a substep pays.
// Also: the same chain as ONE captured hipGraph launch per 100 pairs.
//   hipcc --offload-arch=gfx950 -O3 launch_chain.hip -o launch_chain.exe && ./launch_chain.exe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void ka(float4* a, const float4* b) { if (threadIdx.x == 0) a[blockIdx.x] = b[blockIdx.x % 64]; }
__global__ void kb(float4* b, const float4* a) { if (threadIdx.x == 0) b[blockIdx.x] = a[blockIdx.x % 64]; }

static double run(int ga, int gb, int pairs, hipStream_t st, float4* a, float4* b) {
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < pairs; ++i) {
        hipLaunchKernelGGL(ka, dim3(ga), dim3(256), 0, st, a, b);
        hipLaunchKernelGGL(kb, dim3(gb), dim3(64), 0, st, b, a);
    }
    hipStreamSynchronize(st);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / pairs;
}

int main() {
    float4 *a, *b;
    CK(hipMalloc(&a, 8192 * sizeof(float4))); CK(hipMalloc(&b, 8192 * sizeof(float4)));
    CK(hipMemset(a, 0, 8192 * sizeof(float4))); CK(hipMemset(b, 0, 8192 * sizeof(float4)));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int geo[2][2] = {{526, 576}, {5557, 3744}};
    for (auto& g : geo) {
        run(g[0], g[1], 200, st, a, b);
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) best = std::min(best, run(g[0], g[1], 2000, st, a, b));
        // graph of 100 pairs
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 100; ++i) {
            hipLaunchKernelGGL(ka, dim3(g[0]), dim3(256), 0, st, a, b);
            hipLaunchKernelGGL(kb, dim3(g[1]), dim3(64), 0, st, b, a);
        }
        CK(hipStreamEndCapture(st, &graph)); CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        CK(hipGraphLaunch(exec, st)); CK(hipStreamSynchronize(st));
        double gbest = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 20; ++k) CK(hipGraphLaunch(exec, st));
            CK(hipStreamSynchronize(st));
            gbest = std::min(gbest, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000.0);
        }
        printf("empty producer/consumer pair, grids %d x 256 + %d x 64: %.2f us per pair on a stream, %.2f us per pair inside a captured graph\n",
               g[0], g[1], best, gbest);
        hipGraphExecDestroy(exec); hipGraphDestroy(graph);
    }
    return 0;
}
