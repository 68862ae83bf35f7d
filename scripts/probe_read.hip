// Read-only streaming ceiling on this part: every lane keeps U 16-byte loads in flight, grid-stride over 4 GiB.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int U>
__global__ __launch_bounds__(256) void rd(const float4 *x, size_t n, float *out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int U>
void run(const float4 *x, size_t n, float *out, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(rd<U>, dim3(blocks), dim3(256), 0, 0, x, n, out);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(rd<U>, dim3(blocks), dim3(256), 0, 0, x, n, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("U=%d blocks=%5d: %.2f TB/s\n", U, blocks, n * 16.0 / (ms / 10 * 1e-3) / 1e12);
}
int main() {
    const size_t bytes = 4ull << 30, n = bytes / 16;
    float4 *x; float *out; hipMalloc(&x, bytes); hipMalloc(&out, 4); hipMemset(x, 1, bytes);
    for (int blocks : {1024, 2048, 4096, 8192}) { run<4>(x, n, out, blocks); run<8>(x, n, out, blocks); run<16>(x, n, out, blocks); }
    return 0;
}
