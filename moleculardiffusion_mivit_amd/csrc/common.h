// Shared device/host helpers for the MiViT gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string>
#include <functional>

#include "../../include/mivit_hip.h"

// ------------------------------------------------------------------------------------------------
// error plumbing (thread-local message, C-ABI returns an int)
// ------------------------------------------------------------------------------------------------
void mivit_set_error(const char *fmt, ...);
#define MIVIT_FAIL(...)                 \
    do {                                \
        mivit_set_error(__VA_ARGS__);   \
        return 1;                       \
    } while (0)
#define MIVIT_CHECK(cond, ...)          \
    do {                                \
        if (!(cond)) MIVIT_FAIL(__VA_ARGS__); \
    } while (0)
#define MIVIT_HIP(call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) MIVIT_FAIL("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define MIVIT_LAUNCH_CHECK()                                                                  \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) MIVIT_FAIL("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// element types
// ------------------------------------------------------------------------------------------------
// 16-bit storage types: KIND 0 = bfloat16 (the fast path), 1 = IEEE half (fp16 mode, general kernels only)
template <int KIND>
struct h16 {
    uint16_t v;
};
typedef h16<0> bf16;
typedef h16<1> f16;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return __uint_as_float(((uint32_t)x.v) << 16); }
__device__ __forceinline__ float to_f32(f16 x) { return (float)__builtin_bit_cast(_Float16, x.v); }

template <typename T>
__device__ __forceinline__ T from_f32(float x);
template <>
__device__ __forceinline__ float from_f32<float>(float x) {
    return x;
}
template <>
__device__ __forceinline__ bf16 from_f32<bf16>(float x) {
    __bf16 b = (__bf16)x;  // RNE, NaN-preserving (v_cvt_pk_bf16_f32)
    bf16 r;
    r.v = __builtin_bit_cast(unsigned short, b);
    return r;
}

static inline size_t dtype_size(int dtype) { return dtype == MIVIT_F32 ? 4 : 2; }

// activation and its derivative.  `saved` is the post-activation for relu/leaky, the PRE-activation for gelu.
__device__ __forceinline__ float act_fwd(int act, float u) {
    switch (act) {
        case MIVIT_ACT_RELU: return u > 0.f ? u : 0.f;
        case MIVIT_ACT_LEAKY_RELU: return u > 0.f ? u : 0.01f * u;
        case MIVIT_ACT_GELU: return 0.5f * u * (1.f + erff(u * 0.70710678118654752f));
        default: return u;
    }
}
__device__ __forceinline__ float act_bwd(int act, float saved) {
    switch (act) {
        case MIVIT_ACT_RELU: return saved > 0.f ? 1.f : 0.f;
        case MIVIT_ACT_LEAKY_RELU: return saved > 0.f ? 1.f : 0.01f;
        case MIVIT_ACT_GELU: {
            const float u = saved;
            const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * u * u);
            return cdf + u * pdf;
        }
        default: return 1.f;
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA traits: one 16x16 output tile per wave-instruction.
//   C/D (both dtypes): lane l holds column (l & 15), rows (l >> 4) * 4 + j, j = 0..3.
//   f32  (v_mfma_f32_16x16x4_f32):   A[row = l&15][k = l>>4],           B[k = l>>4][col = l&15]
//   bf16 (v_mfma_f32_16x16x32_bf16): A[row = l&15][k = 8*(l>>4) + j],   B[k = 8*(l>>4) + j][col = l&15], j = 0..7
// Operand images in LDS are [row-or-col][k]; the bf16 fragment is one 16-byte read (k contiguous),
// the f32 fragment a single dword (any k stride).
// ------------------------------------------------------------------------------------------------
template <typename T>
struct Mma;

template <>
struct Mma<float> {
    static constexpr int KS = 4;
    typedef float Frag;
    static __device__ __forceinline__ Frag load(const float *img, int rs, int ks, int r0, int k0, int lane) {
        return img[(r0 + (lane & 15)) * rs + (k0 + (lane >> 4)) * ks];
    }
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};

template <>
struct Mma<bf16> {
    static constexpr int KS = 32;
    typedef bf16x8 Frag;
    // ks must be 1; (r0 + row) * rs + k0 + 8 * (lane >> 4) must be a multiple of 8 elements (16 bytes)
    static __device__ __forceinline__ Frag load(const bf16 *img, int rs, int /*ks*/, int r0, int k0, int lane) {
        return *reinterpret_cast<const bf16x8 *>(img + (r0 + (lane & 15)) * rs + k0 + 8 * (lane >> 4));
    }
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

template <>
struct Mma<f16> {
    static constexpr int KS = 32;
    typedef f16x8 Frag;
    static __device__ __forceinline__ Frag load(const f16 *img, int rs, int /*ks*/, int r0, int k0, int lane) {
        return *reinterpret_cast<const f16x8 *>(img + (r0 + (lane & 15)) * rs + k0 + 8 * (lane >> 4));
    }
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

// Cross-lane reductions without the LDS crossbar.  __shfl_xor compiles to ds_bpermute_b32 (an LDS-pipe round trip per
// step); inside a 16-lane row a DPP row rotation does the same job as a modifier of the VALU add itself, and gfx950's
// v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane rows between two registers (both operands = v: afterwards
// one holds the even rows twice, the other the odd rows twice, and their sum / max is the xor-16 / xor-32 butterfly step).
// Checked lane by lane against plain loops by scripts/probe_dpp.hip.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// all 16 lanes of a row (lanes 16r .. 16r+15) get the row's sum / max            (row_ror:8, 4, 2, 1)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0x128>(v); v += dpp_mov<0x124>(v); v += dpp_mov<0x122>(v); return v + dpp_mov<0x121>(v);
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0x128>(v)); v = fmaxf(v, dpp_mov<0x124>(v)); v = fmaxf(v, dpp_mov<0x122>(v)); return fmaxf(v, dpp_mov<0x121>(v));
}
// across the four rows (lanes l, l ^ 16, l ^ 32, l ^ 48): every lane gets the sum / max.  The instruction is written out:
// with __builtin_amdgcn_permlane{16,32}_swap this compiler (ROCm 7.2) hands back result[1] in the register of result[0]
// (scripts/probe_dpp.hip: every lane got 4 x its own value).  The s_nop pads cover the VALU -> permlane hazards the
// compiler does not see through inline asm.
__device__ __forceinline__ void rows_swap32(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void rows_swap16(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float rows4_sum(float v) {
    float a = v, b = v;
    rows_swap32(a, b);
    float t = a + b, c = t;
    rows_swap16(t, c);
    return t + c;
}
__device__ __forceinline__ float rows4_max(float v) {
    float a = v, b = v;
    rows_swap32(a, b);
    float t = fmaxf(a, b), c = t;
    rows_swap16(t, c);
    return fmaxf(t, c);
}

// wave-level helpers (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 16-byte vector access of V = 16 / sizeof(T) consecutive elements, widened to fp32
template <>
__device__ __forceinline__ f16 from_f32<f16>(float x) {
    f16 r;
    r.v = __builtin_bit_cast(unsigned short, (_Float16)x);   // RNE; overflow -> inf (what loss scaling watches for)
    return r;
}
__device__ __forceinline__ void load16(const float *p, float *out) {
    const float4 v = *reinterpret_cast<const float4 *>(p);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
__device__ __forceinline__ void load16(const bf16 *p, float *out) {
    const uint4 v = *reinterpret_cast<const uint4 *>(p);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        out[2 * i] = __uint_as_float(w[i] << 16);
        out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void load16(const f16 *p, float *out) {
    const f16x8 v = *reinterpret_cast<const f16x8 *>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
}
__device__ __forceinline__ void store16(f16 *p, const float *v) {
    f16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (_Float16)v[i];
    *reinterpret_cast<f16x8 *>(p) = o;
}
__device__ __forceinline__ void store16(float *p, const float *v) {
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store16(bf16 *p, const float *v) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        w[i] = (uint32_t)from_f32<bf16>(v[2 * i]).v | ((uint32_t)from_f32<bf16>(v[2 * i + 1]).v << 16);
    *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// internal launchers shared between the op-level C-ABI and the engine (all return 0 / non-zero)
struct LinearFwdArgs {
    int dtype;
    const void *x; int x_is_f32; int64_t ldx;
    const void *W; int w_is_bf16;   // fp32 master weights, or (bf16 mode) the per-step bf16 shadow copy
    const float *bias;
    int M, N, K, act;
    const void *resid; int64_t ldr;
    void *y; int64_t ldy; void *y_preact;
    // optional output row map: out_row = (m / map_rows) * map_stride + m % map_rows + map_off  (map_rows > 0)
    int map_rows, map_stride, map_off;
    int y_is_f32;   // write fp32 output regardless of dtype (model output)
};
int launch_linear_fwd(const LinearFwdArgs &a, hipStream_t s);

struct LinearDgradArgs {
    int dtype;
    const void *dy; int dy_is_f32; int64_t lddy;
    const void *W; int w_is_bf16;
    int M, N, K;          // dy [M,N], W [N,K], dx [M,K]
    int act; const void *saved; int64_t lds;
    const void *dres; int64_t lddr;
    void *dx; int64_t lddx; int dx_is_f32;
};
int launch_linear_dgrad(const LinearDgradArgs &a, hipStream_t s);

struct LinearWgradArgs {
    int dtype;
    const void *dy; int dy_is_f32; int64_t lddy;
    const void *x; int x_is_f32; int64_t ldx;
    int M, N, K;          // dW [N,K] = dy^T x ; db [N]
    float *dW; float *db; int accumulate;
    void *ws; size_t ws_bytes;
};
size_t linear_wgrad_ws_bytes(int M, int N, int K);
int launch_linear_wgrad(const LinearWgradArgs &a, hipStream_t s);

struct LayerNormFwdArgs {
    int dtype;
    const void *z; int64_t ldz; const float *gamma; const float *beta;
    int M, E;
    void *y; int64_t ldy; int rows_per_seq, out_seq_stride, out_row_off;
    const float *pos; float *mean; float *rstd;
    // optional input row map (gather): in_row = (r / in_rows) * in_stride + r % in_rows + in_off  (in_rows > 0)
    int in_rows, in_stride, in_off;
};
int launch_layernorm_fwd(const LayerNormFwdArgs &a, hipStream_t s);

struct LayerNormBwdArgs {
    int dtype;
    const void *dy; int64_t lddy; const void *z; int64_t ldz;
    const float *gamma; const float *mean; const float *rstd;
    int M, E;
    int rows_per_seq, in_seq_stride, in_row_off;   // row map of dy (as written by the forward)
    int z_rows, z_stride, z_off;                   // row map of z  (gather used by the forward), z_rows > 0
    void *dz; int64_t lddz;                        // written through the z row map
    float *dgamma; float *dbeta; int accumulate;
    float *dzsum;                                  // optional: column sums of dz (= bias gradient of the Linear whose
                                                   // output fed this LayerNorm through the residual add)
    void *ws; size_t ws_bytes;
};
size_t layernorm_bwd_ws_bytes(int M, int E);
int launch_layernorm_bwd(const LayerNormBwdArgs &a, hipStream_t s);

int launch_attention_fwd(int dtype, const void *qkv, int B, int S, int H, int Dh, void *ctx, hipStream_t s);
int launch_attention_bwd(int dtype, const void *qkv, const void *dctx, int B, int S, int H, int Dh, void *dqkv,
                         hipStream_t s);
int attention_max_seq(int dtype, int Dh);
// bf16 short-sequence fast path (attention_fast.hip)
bool attention_fast_supported(int dtype, int S, int Dh);
int launch_attention_fwd_fast(const void *qkv, int B, int S, int H, int Dh, void *ctx, hipStream_t s);
int launch_attention_bwd_fast(const void *qkv, const void *dctx, int B, int S, int H, int Dh, void *dqkv, hipStream_t s);

// LDS-DMA streaming kernels of the frame embedding, bf16 mode (embed.hip)
bool embed_dma_supported(int dtype, int M, int K, int E);
int launch_embed_fwd_dma(const float *X, const void *W_bf16, const float *bias, void *Y, int M, int K, int E, hipStream_t s);
size_t embed_wgrad_dma_ws_bytes(int M, int K, int E);
int launch_embed_wgrad_dma(const void *dY_bf16, const float *X, float *dW, int M, int K, int E, void *ws, size_t ws_bytes,
                           hipStream_t s);

// bf16 row-stream GEMM with resident weights (rowstream.hip)
bool rowstream_supported(int M, int N, int K, bool dgrad, int64_t lda, int64_t ldw, const void *A, const void *W);
int launch_rowstream(bool dgrad, const void *A, int64_t lda, const void *W_bf16, int64_t ldw, int M, int N, int K,
                     const float *bias, int act, const void *dact, int64_t ldd, int dact_kind, const void *resid,
                     int64_t ldr, void *Cout, int64_t ldc, void *C2, const float *gamma, const float *beta, void *Y,
                     int64_t ldy, float *mean, float *rstd, hipStream_t s);

// bf16 wave-stream GEMM (wavestream.hip): same contract as launch_rowstream, K in {128, 256}
bool wavestream_supported(int M, int N, int K, bool dgrad, int64_t lda, int64_t ldw, const void *A, const void *W);
int launch_wavestream(bool dgrad, const void *A, int64_t lda, const void *W_bf16, int64_t ldw, int M, int N, int K,
                      const float *bias, int act, const void *dact, int64_t ldd, int dact_kind, const void *resid,
                      int64_t ldr, void *Cout, int64_t ldc, void *C2, const float *gamma, const float *beta, void *Y,
                      int64_t ldy, float *mean, float *rstd, hipStream_t s);

// bf16 LDS-DMA GEMMs of wide layers (gemm_dma.hip): K / N beyond the row-stream kernels' resident-weight budget
bool gemm_dma_supported(int M, int NC, int KC, bool dgrad);
int launch_gemm_dma_fwd(const void *x, int64_t ldx, const void *W_bf16, const float *bias, int M, int N, int K, int act,
                        const void *resid, int64_t ldr, void *y, int64_t ldy, void *y_preact, hipStream_t s);
int launch_gemm_dma_dgrad(const void *dy, int64_t lddy, const void *W_bf16, int M, int N, int K, int act, const void *saved,
                          int64_t lds, const void *dres, int64_t lddr, void *dx, int64_t lddx, hipStream_t s);

// bf16 LDS-DMA weight gradient of the layer projections (wgrad_dma.hip)
bool wgrad_dma_supported(int M, int N, int K, int64_t lddy, int64_t ldx, const void *dy, const void *x);
size_t wgrad_dma_ws_bytes(int M, int N, int K);
int launch_wgrad_dma(const void *dy, int64_t lddy, const void *x, int64_t ldx, int M, int N, int K, float *dW, float *db,
                     void *ws, size_t ws_bytes, hipStream_t s);

// bf16 weight gradient of narrow layers (wgrad_small.hip): (N, K) in {(64,64), (128,64), (192,64), (64,128)}
bool wgrad_small_supported(int M, int N, int K, int64_t lddy, int64_t ldx, const void *dy, const void *x);
size_t wgrad_small_ws_bytes(int M, int N, int K);
int launch_wgrad_small(const void *dy, int64_t lddy, const void *x, int64_t ldx, int M, int N, int K, float *dW, float *db,
                       void *ws, size_t ws_bytes, hipStream_t s);


// linear frame embedding of SMALL frames (patch sizes up to 16 x 16, any row length: wavestream.hip / wgrad_small.hip, AF32 / XF32)
bool embed_small_fwd_supported(int M, int K, int E, const void *X, const void *Y);
int launch_embed_small_fwd(const float *X, const void *W_bf16, const float *bias, void *Y, int M, int K, int E, hipStream_t s);
bool embed_small_wgrad_supported(int M, int N, int K, int64_t lddy, const void *dy, const void *x);
size_t embed_small_wgrad_ws_bytes(int M, int N, int K);
int launch_embed_small_wgrad(const void *dy, int64_t lddy, const float *x, int M, int N, int K, float *dW, float *db, void *ws,
                             size_t ws_bytes, hipStream_t s);

// ... and their IEEE-half builds
bool embed_small_fwd_supported_f16(int M, int K, int E, const void *X, const void *Y);
int launch_embed_small_fwd_f16(const float *X, const void *W_bf16, const float *bias, void *Y, int M, int K, int E, hipStream_t s);
bool embed_small_wgrad_supported_f16(int M, int N, int K, int64_t lddy, const void *dy, const void *x);
size_t embed_small_wgrad_ws_bytes_f16(int M, int N, int K);
int launch_embed_small_wgrad_f16(const void *dy, int64_t lddy, const float *x, int M, int N, int K, float *dW, float *db, void *ws,
                             size_t ws_bytes, hipStream_t s);

// the same seven units compiled with -DMIVIT_ELEM_F16 (elem.h): IEEE-half operands and stored activations, same contracts
bool attention_fast_supported_f16(int dtype, int S, int Dh);
int launch_attention_fwd_fast_f16(const void *qkv, int B, int S, int H, int Dh, void *ctx, hipStream_t s);
int launch_attention_bwd_fast_f16(const void *qkv, const void *dctx, int B, int S, int H, int Dh, void *dqkv, hipStream_t s);
bool embed_dma_supported_f16(int dtype, int M, int K, int E);
int launch_embed_fwd_dma_f16(const float *X, const void *W_bf16, const float *bias, void *Y, int M, int K, int E, hipStream_t s);
size_t embed_wgrad_dma_ws_bytes_f16(int M, int K, int E);
int launch_embed_wgrad_dma_f16(const void *dY_bf16, const float *X, float *dW, int M, int K, int E, void *ws, size_t ws_bytes,
                           hipStream_t s);
bool rowstream_supported_f16(int M, int N, int K, bool dgrad, int64_t lda, int64_t ldw, const void *A, const void *W);
int launch_rowstream_f16(bool dgrad, const void *A, int64_t lda, const void *W_bf16, int64_t ldw, int M, int N, int K,
                     const float *bias, int act, const void *dact, int64_t ldd, int dact_kind, const void *resid,
                     int64_t ldr, void *Cout, int64_t ldc, void *C2, const float *gamma, const float *beta, void *Y,
                     int64_t ldy, float *mean, float *rstd, hipStream_t s);
bool wavestream_supported_f16(int M, int N, int K, bool dgrad, int64_t lda, int64_t ldw, const void *A, const void *W);
int launch_wavestream_f16(bool dgrad, const void *A, int64_t lda, const void *W_bf16, int64_t ldw, int M, int N, int K,
                      const float *bias, int act, const void *dact, int64_t ldd, int dact_kind, const void *resid,
                      int64_t ldr, void *Cout, int64_t ldc, void *C2, const float *gamma, const float *beta, void *Y,
                      int64_t ldy, float *mean, float *rstd, hipStream_t s);
bool wgrad_dma_supported_f16(int M, int N, int K, int64_t lddy, int64_t ldx, const void *dy, const void *x);
size_t wgrad_dma_ws_bytes_f16(int M, int N, int K);
int launch_wgrad_dma_f16(const void *dy, int64_t lddy, const void *x, int64_t ldx, int M, int N, int K, float *dW, float *db,
                     void *ws, size_t ws_bytes, hipStream_t s);
bool wgrad_small_supported_f16(int M, int N, int K, int64_t lddy, int64_t ldx, const void *dy, const void *x);
size_t wgrad_small_ws_bytes_f16(int M, int N, int K);
int launch_wgrad_small_f16(const void *dy, int64_t lddy, const void *x, int64_t ldx, int M, int N, int K, float *dW, float *db,
                       void *ws, size_t ws_bytes, hipStream_t s);

// small elementwise / reduction helpers (misc.hip)
// out[n] (+)= sum_p part[p*n_stride + n]   (deterministic slab reduce)
int launch_slab_reduce(const float *part, int nparts, int64_t n, float *out, int accumulate, hipStream_t s);
int launch_slab_reduce_strided(const float *part, int nparts, int64_t stride, int64_t n, float *out, hipStream_t s);
// queue the strided reductions issued from here on (their slabs must stay untouched) / run the queue as one launch
void slab_defer_begin();
void slab_defer_cancel();
int slab_defer_flush(hipStream_t s);
// up to three (slab, output) pairs of the same shape in one launch; pass null outputs from the end
int launch_slab_reduce3(const float *p0, float *o0, const float *p1, float *o1, const float *p2, float *o2, int nparts,
                        int n, int accumulate, hipStream_t s);
// tokens[b, 0, :] = reg[:] (+ add[b, :]) (+ pos[0, :])   -- regression token row (models.py:339-347)
int launch_reg_token_fill(int dtype, void *tokens, int B, int S, int E, const float *reg, const void *add,
                          const float *pos, hipStream_t s);
// mean over tokens: out[b,:] = mean_s x[b,s,:]  and its backward (models.py:354)
int launch_mean_pool_fwd(int dtype, const void *x, int B, int S, int E, void *out, hipStream_t s);
int launch_mean_pool_bwd(int dtype, const void *dout, int B, int S, int E, void *dx, hipStream_t s);
// column sums over the batch of selected token rows: out[s_sel, e] = sum_b x[b, s0 + s_sel, e]
// (gradient of the positional table and of the regression token).  ws: B-chunk partials.
size_t batch_colsum_ws_bytes(int B, int rows, int E);
int launch_batch_colsum(int dtype, const void *x, int B, int S, int E, int s0, int rows, float *out, void *ws,
                        size_t ws_bytes, hipStream_t s);
// generic conversions / copies
// 16-bit sides are of kind `dtype16` (MIVIT_BF16 or MIVIT_F16)
int launch_convert(int src_is_f32, const void *src, int64_t lds_, int dst_dtype_is_f32, void *dst, int64_t ldd,
                   int rows, int cols, int accumulate, hipStream_t s, int dtype16 = MIVIT_BF16);
int launch_fill_zero(void *p, size_t bytes, hipStream_t s);
// dW[n][k] = dW[n][k] * gamma[k] + db[n] * beta[k]  (weight gradient taken against normalised inputs, see misc.hip)
int launch_affine_fixup(float *dW, const float *db, const float *gamma, const float *beta, int N, int K, hipStream_t s);

// in-library kernel timing (misc.hip): the engine sets the category, launch sites bracket their main kernel
void prof_set_tag(int tag);
void prof_pin_tag(int tag);          // >= 0: every prof_set_tag until prof_pin_tag(-1) resolves to this tag
bool prof_begin(hipStream_t s);
void prof_end(hipStream_t s);
struct ProfScope {
    hipStream_t s; bool on;
    explicit ProfScope(hipStream_t st) : s(st), on(prof_begin(st)) {}
    ~ProfScope() { if (on) prof_end(s); }
};

// hipGraph replay of launch-bound call sequences (misc.hip).  `key` must hold every value the body's launches depend on
// (pointers, sizes).  First sighting of a key: body(s) runs directly.  Second: body is captured on an internal stream,
// instantiated and launched on s.  Later: replay.  Disabled by MIVIT_GRAPHS=0, while the in-library profiler is on, and
// after repeated capture failures; small LRU.
int graph_run(const uint64_t *key, int nkey, hipStream_t s, const std::function<int(hipStream_t)> &body);
