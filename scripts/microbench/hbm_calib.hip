// Calibration kernels for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section): known byte
// counts moved with the access widths the product kernels use (4 B and 16 B per lane, coalesced), past the 256 MiB
// Infinity Cache (2 GiB buffers), so that counter-to-byte factors can be applied to the kernels' own readings.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void calib_read4(const float* __restrict__ x, float* __restrict__ out, size_t n) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += x[i];
    if (s == 12345.678f) out[0] = s;
}
__global__ void calib_read16(const float4* __restrict__ x, float* __restrict__ out, size_t n4) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const float4 v = x[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ void calib_write4(float* __restrict__ x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = 1.0f;
}
__global__ void calib_write16(float4* __restrict__ x, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) x[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main() {
    const size_t bytes = 2ull << 30;  // 2 GiB
    float *a, *out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(a, 0, bytes));
    const size_t n = bytes / 4;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(calib_read4, dim3(8192), dim3(256), 0, 0, a, out, n);
        hipLaunchKernelGGL(calib_read16, dim3(8192), dim3(256), 0, 0, (const float4*)a, out, n / 4);
        hipLaunchKernelGGL(calib_write4, dim3(8192), dim3(256), 0, 0, a, n);
        hipLaunchKernelGGL(calib_write16, dim3(8192), dim3(256), 0, 0, (float4*)a, n / 4);
    }
    CK(hipDeviceSynchronize());
    printf("calibration kernels moved %zu bytes each\n", bytes);
    return 0;
}
