// How fast can the frame-embedding forward's access pattern be read?  X [M][K] fp32, K = 4096 (16 KB rows).  A wave owns
// 32 rows (2 tiles of 16); lane (cq, g) reads row cq at k = 4g .. of every 16-float group, as the MFMA operand wants it.
// RUN = floats of a row fetched back to back (64: one k stage per iteration, the kernel today; 256: four stages at once).
// stag = per-workgroup rotation of the k loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int RUN>
__global__ __launch_bounds__(256) void rows(const float *x, int M, int K, int stag, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, cq = lane & 15;
    const int m0 = blockIdx.x * 128 + wave * 32;
    const float *p0 = x + (size_t)(m0 + cq) * K + 4 * g, *p1 = p0 + (size_t)16 * K;
    const int nit = K / RUN, s0 = (int)(((unsigned)blockIdx.x * (unsigned)stag) % (unsigned)nit);
    float acc = 0.f;
    float4 nx[2][RUN / 16], cx[2][RUN / 16];
    auto load = [&](int it, float4 (&d)[2][RUN / 16]) {
        int t = it + s0; t = t >= nit ? t - nit : t;
#pragma unroll
        for (int u = 0; u < RUN / 16; ++u) { d[0][u] = *reinterpret_cast<const float4 *>(p0 + t * RUN + 16 * u); d[1][u] = *reinterpret_cast<const float4 *>(p1 + t * RUN + 16 * u); }
    };
    load(0, nx);
    for (int it = 0; it < nit; ++it) {
#pragma unroll
        for (int u = 0; u < RUN / 16; ++u) { cx[0][u] = nx[0][u]; cx[1][u] = nx[1][u]; }
        load(it + 1 < nit ? it + 1 : it, nx);
#pragma unroll
        for (int u = 0; u < RUN / 16; ++u) acc += cx[0][u].x + cx[0][u].w + cx[1][u].y + cx[1][u].z;
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int RUN>
void run(const float *x, int M, int K, int stag, float *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(rows<RUN>, dim3(M / 128), dim3(256), 0, 0, x, M, K, stag, out);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(rows<RUN>, dim3(M / 128), dim3(256), 0, 0, x, M, K, stag, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("run %4d B/row  stag %2d: %.2f TB/s\n", RUN * 4, stag, (double)M * K * 4 / (ms / 5 * 1e-3) / 1e12);
}
int main() {
    const int M = 524288, K = 4096;
    float *x, *out; hipMalloc(&x, (size_t)M * K * 4); hipMalloc(&out, 4); hipMemset(x, 1, (size_t)M * K * 4);
    for (int stag : {0, 17}) { run<64>(x, M, K, stag, out); run<128>(x, M, K, stag, out); run<256>(x, M, K, stag, out); }
    return 0;
}
