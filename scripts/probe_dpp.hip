#include <hip/hip_runtime.h>
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float g16_sum(float v) {
    v += dpp<0x128>(v); v += dpp<0x124>(v); v += dpp<0x122>(v); v += dpp<0x121>(v); return v;
}
__device__ __forceinline__ float g16_max(float v) {
    v = fmaxf(v, dpp<0x128>(v)); v = fmaxf(v, dpp<0x124>(v)); v = fmaxf(v, dpp<0x122>(v)); v = fmaxf(v, dpp<0x121>(v)); return v;
}
// both results of the swap builtins come back as the SAME register in this compiler (result[1] aliases result[0]), so the
// instruction is written out: after it, a holds the even rows twice and b the odd rows twice
__device__ __forceinline__ float x4_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    float t = a + b, c = t;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(t), "+v"(c));
    return t + c;
}
__global__ void k(float *o, const float *i) {
    float v = i[threadIdx.x];
    o[threadIdx.x] = g16_sum(v); o[64 + threadIdx.x] = g16_max(v); o[128 + threadIdx.x] = x4_sum(v);
}
int main() {
    float *di, *dout, hi[64], ho[192];
    for (int i = 0; i < 64; ++i) hi[i] = (float)((i * 37) % 11) - 3.f;
    (void)hipMalloc(&di, 256); (void)hipMalloc(&dout, 768); (void)hipMemcpy(di, hi, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, di);
    (void)hipMemcpy(ho, dout, 768, hipMemcpyDeviceToHost);
    int bad[3] = {0, 0, 0};
    for (int i = 0; i < 64; ++i) {
        float s = 0, m = -1e30f, x = 0;
        for (int j = 0; j < 16; ++j) { s += hi[(i & 48) + j]; m = fmaxf(m, hi[(i & 48) + j]); }
        for (int g = 0; g < 4; ++g) x += hi[(i & 15) + 16 * g];
        if (ho[i] != s) { if (bad[0]++ < 4) printf("row16_sum lane %d: got %f want %f\n", i, ho[i], s); }
        if (ho[64 + i] != m) { if (bad[1]++ < 4) printf("row16_max lane %d: got %f want %f\n", i, ho[64 + i], m); }
        if (ho[128 + i] != x) { if (bad[2]++ < 4) printf("rows4_sum lane %d: got %f want %f\n", i, ho[128 + i], x); }
    }
    printf("bad: row16_sum %d row16_max %d rows4_sum %d\n", bad[0], bad[1], bad[2]);
    return bad[0] + bad[1] + bad[2] != 0;
}
