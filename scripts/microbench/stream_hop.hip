// What does a dependency edge BETWEEN two streams cost on this part?  The intra-scene overlap of the MPM grid kernel
// lives or dies by it: a chain  A(s1) -> event -> B(s2) -> event -> A(s1) ...  of near-empty kernels, against the same chain on ONE
// stream, and against two INDEPENDENT chains on two streams (what two scenes per GPU do).  Also a "busy" variant where kernel A runs
// ~20 us, to see whether the hop hides under a running kernel of the waiting stream.
//   hipcc --offload-arch=gfx950 -O3 stream_hop.hip -o stream_hop.exe && ./stream_hop.exe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void tiny(float4* a, const float4* b) { if (threadIdx.x == 0) a[blockIdx.x] = b[blockIdx.x % 64]; }
__global__ void spin(float4* a, const float4* b, long cycles) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (threadIdx.x == 0) a[blockIdx.x] = b[blockIdx.x % 64];
}
using clk = std::chrono::steady_clock;
static double us(clk::time_point t0) { return std::chrono::duration<double, std::micro>(clk::now() - t0).count(); }

int main() {
    float4 *a, *b, *c, *d;
    CK(hipMalloc(&a, 8192 * 16)); CK(hipMalloc(&b, 8192 * 16)); CK(hipMalloc(&c, 8192 * 16)); CK(hipMalloc(&d, 8192 * 16));
    CK(hipMemset(a, 0, 8192 * 16)); CK(hipMemset(b, 0, 8192 * 16)); CK(hipMemset(c, 0, 8192 * 16)); CK(hipMemset(d, 0, 8192 * 16));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int N = 1000;
    hipEvent_t ev[2 * N];
    for (auto& evt : ev) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
    const int GA = 2200, GB = 3744;     // half a 1 M block kernel's work list; the grid kernel's active blocks
    for (int rep = 0; rep < 3; ++rep) {
        // (1) one stream: A B A B ...
        CK(hipDeviceSynchronize());
        auto t0 = clk::now();
        for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny, dim3(GA), dim3(256), 0, s1, a, b); hipLaunchKernelGGL(tiny, dim3(GB), dim3(64), 0, s1, b, a); }
        CK(hipDeviceSynchronize());
        const double one = us(t0) / N;
        // (2) two streams, every edge crosses: A(s1) -> B(s2) -> A(s1) ...
        t0 = clk::now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(tiny, dim3(GA), dim3(256), 0, s1, a, b);
            CK(hipEventRecord(ev[2 * i], s1)); CK(hipStreamWaitEvent(s2, ev[2 * i], 0));
            hipLaunchKernelGGL(tiny, dim3(GB), dim3(64), 0, s2, b, a);
            CK(hipEventRecord(ev[2 * i + 1], s2)); CK(hipStreamWaitEvent(s1, ev[2 * i + 1], 0));
        }
        CK(hipDeviceSynchronize());
        const double cross = us(t0) / N;
        // (3) two independent chains
        t0 = clk::now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(tiny, dim3(GA), dim3(256), 0, s1, a, b); hipLaunchKernelGGL(tiny, dim3(GB), dim3(64), 0, s1, b, a);
            hipLaunchKernelGGL(tiny, dim3(GA), dim3(256), 0, s2, c, d); hipLaunchKernelGGL(tiny, dim3(GB), dim3(64), 0, s2, d, c);
        }
        CK(hipDeviceSynchronize());
        const double indep = us(t0) / N;
        // (4) the overlap pattern itself with 25-us block halves: s1: A1 A2 | s2: G1 after A1 (under A2), G2 after A2; next A1 waits G1+G2
        const long cyc = 2500;   // wall_clock64 runs at 100 MHz: 25 us
        t0 = clk::now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, a, b, cyc);
            CK(hipEventRecord(ev[2 * i], s1)); CK(hipStreamWaitEvent(s2, ev[2 * i], 0));
            hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s2, c, a, (long)500);          // G1: 5 us, under A2
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, b, a, cyc);               // A2
            CK(hipEventRecord(ev[2 * i + 1], s1)); CK(hipStreamWaitEvent(s2, ev[2 * i + 1], 0));
            hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s2, d, b, (long)200);          // G2 (seam): 2 us, exposed
            CK(hipEventRecord(ev[2 * i], s2)); CK(hipStreamWaitEvent(s1, ev[2 * i], 0));
        }
        CK(hipDeviceSynchronize());
        const double pat = us(t0) / N;
        // (5) the same work on one stream: A1 A2 G(7 us)
        t0 = clk::now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, a, b, cyc);
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, b, a, cyc);
            hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s1, c, a, (long)700);
        }
        CK(hipDeviceSynchronize());
        const double ser = us(t0) / N;
        // (6) the overlap pattern as a captured graph (fork / join through events while capturing: edges become graph dependencies)
        double gpat = -1.0;
        {
            const int M = 50;
            hipGraph_t graph; hipGraphExec_t exec;
            CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < M; ++i) {
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, a, b, cyc);
                CK(hipEventRecord(ev[2 * i], s1)); CK(hipStreamWaitEvent(s2, ev[2 * i], 0));
                hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s2, c, a, (long)500);
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, b, a, cyc);
                CK(hipEventRecord(ev[2 * i + 1], s1)); CK(hipStreamWaitEvent(s2, ev[2 * i + 1], 0));
                hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s2, d, b, (long)200);
                CK(hipEventRecord(ev[2 * i], s2)); CK(hipStreamWaitEvent(s1, ev[2 * i], 0));
            }
            CK(hipStreamEndCapture(s1, &graph)); CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            CK(hipGraphLaunch(exec, s1)); CK(hipStreamSynchronize(s1));
            t0 = clk::now();
            for (int k = 0; k < 10; ++k) CK(hipGraphLaunch(exec, s1));
            CK(hipStreamSynchronize(s1));
            gpat = us(t0) / (10.0 * M);
            hipGraphExecDestroy(exec); hipGraphDestroy(graph);
        }
        printf("the same pattern as ONE captured graph of 50 substeps (edges = graph dependencies): %.2f us per substep\n", gpat);
        printf("pair of near-empty kernels: one stream %.2f us | every edge across two streams %.2f us (= %.2f us per cross-stream hop more) | two independent chains %.2f us per pair of pairs\n",
               one, cross, (cross - one) / 2, indep);
        printf("overlap pattern (2 x 25 us block halves, 5 us grid half hidden, 2 us seam exposed): %.2f us per substep against %.2f us serial (25 + 25 + 7): %+.2f us\n", pat, ser, pat - ser);
    }
    return 0;
}
