// Probe: __builtin_amdgcn_global_load_lds semantics (16-byte), and counted vmcnt + raw barrier usage.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void probe(const float *src, float *out) {
    __shared__ __attribute__((aligned(16))) float lds[2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each wave DMA-loads 1 KiB: lane l fetches 16 bytes from src + (perm(l)) * 4 floats, lands at ldsbase + l*16
    const int perm = lane ^ 5;
    const float *g = src + wave * 256 + perm * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)(lds + wave * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = lds[i];
}
int main() {
    float *h = (float *)malloc(4096), *d, *o;
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    hipMalloc(&d, 4096); hipMalloc(&o, 4096);
    hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, o);
    hipMemcpy(h, o, 4096, hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) { for (int i = 0; i < 32; ++i) printf("%4.0f ", h[w * 256 + i]); printf("\n"); }
    return 0;
}
