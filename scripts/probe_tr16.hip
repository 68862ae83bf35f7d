// Probe: what does ds_read_b64_tr_b16 deliver?  LDS holds img[row][col] = row*100 + col (16-bit), 64 rows x 64 cols.
// Each lane supplies the address of (row = base_row(group) + q, col = 4p) with i = lane&15 = 4q+p; prints what it got.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(short *out) {
    __shared__ __attribute__((aligned(16))) short img[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i] = (short)((i / 64) * 100 + (i % 64));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = 8 * g + q, col = 16 + 4 * p;     // block: rows 8g..8g+3, cols 16..31
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(&img[row * 64 + col]));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
    short *d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d (g%d i%2d): %5d %5d %5d %5d\n", l, l >> 4, l & 15, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
