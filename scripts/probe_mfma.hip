// What does the matrix pipe deliver with nothing else going on?  Each wave loops over NACC independent 16x16x32 bf16 MFMAs
// on register operands.  Prints TFLOP/s for 1, 2 and 4 waves per SIMD (calibration for the GEMM kernels' roofline).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters) {
    f32x4 acc[NACC];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)(float)(threadIdx.x & 1); }
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) t += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (t == 1234.5f) out[0] = t;
}
template <int NACC>
void run(int blocks_per_cu, int iters, float *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 16 * 16 * 32 * (double)NACC * iters * 4 * grid;
    printf("NACC %2d  workgroups/CU %d (waves/SIMD %d): %.3f ms  %.0f TFLOP/s\n", NACC, blocks_per_cu, blocks_per_cu, ms, fl / ms / 1e9);
}
int main() {
    float *out; hipMalloc(&out, 4);
    run<16>(1, 20000, out); run<16>(2, 20000, out); run<16>(4, 20000, out);
    run<64>(1, 5000, out); run<4>(1, 40000, out); run<4>(2, 40000, out);
    return 0;
}
