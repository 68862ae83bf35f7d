// "Row-stream" GEMM for the encoder-layer projections in bf16 mode:  C[M, N] = epilogue(A[M, K] . W)  with a huge M
// (all tokens of the batch) and a tiny reduction (K = E or F: 128 / 256).
//
// With 2..4 MFMA k-steps per tile a classic tiled GEMM never reaches a steady state: every tile pays an HBM round
// trip in its prologue.  Here the weight slice of the block's 128 output columns is loaded into LDS ONCE and the
// block then walks over 64-row tiles of A: each tile arrives by LDS-DMA (global_load_lds, no staging registers) into
// a 2-slot ring while the previous tile is on the MFMAs and in its epilogue, so the loop is a continuous stream of
// A rows in and C rows out -- an HBM-bound kernel by construction.
//   forward  (W = [N, K], k contiguous): B fragments are 16-byte reads of the natural [n][k] image;
//   dgrad    (W = [K, N], n contiguous): B fragments are ds_read_b64_tr_b16 reads of the natural [k][n] image.
// LDS images are natural row-major tiles whose 16-byte chunks are XOR-permuted on the DMA source address.
// Epilogue (bias, activation, act', residual, optional pre-activation copy) is vectorised through an fp32 staging
// tile; when the block owns whole rows (N == 128) it can also apply the LayerNorm that follows the residual add in
// the post-norm encoder layer (models.py:100-106), writing z, LN(z), mean and rstd in one pass.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

__device__ __forceinline__ void dma16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((gptr_t *)g, (lptr_t *)l, 16, 0, 0);
}
__device__ __forceinline__ void wait_all_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

struct RsArgs {
    const bf16 *A; int64_t lda;
    const bf16 *W; int64_t ldw;
    int M, N, K;
    const float *bias; int act;
    const bf16 *dact; int64_t ldd; int dact_kind;
    const bf16 *resid; int64_t ldr;
    bf16 *C; int64_t ldc;          // forward: the pre-LayerNorm sum z when `gamma` is set
    bf16 *C2;                      // optional pre-activation copy (same ld as C)
    const float *gamma, *beta;     // fused LayerNorm (N == 128 only)
    bf16 *Y; int64_t ldy; float *mean, *rstd;
};

constexpr int BM = 64, BN = 128, NT = 256;

template <int K, bool DGRAD>
struct RsCfg {
    static constexpr int RB = 2 * K;                         // bytes per A row (and per W row in forward)
    static constexpr int A_BYTES = BM * RB;                  // one ring slot
    static constexpr int W_ROWS = DGRAD ? K : BN;
    static constexpr int W_RB = DGRAD ? BN * 2 : RB;
    static constexpr int W_BYTES = W_ROWS * W_RB;
    static constexpr int LDS = W_BYTES + 2 * A_BYTES;
    static constexpr int A_DMA = A_BYTES / 1024 / 4;         // wave-instructions per wave per tile
    static constexpr int W_DMA = W_BYTES / 1024 / 4;
    static_assert(A_BYTES >= 32 * BN * 4, "fp32 staging of 32 rows must fit in one ring slot");
};

template <int K, bool DGRAD>
__device__ __forceinline__ void issue_w(const RsArgs &a, unsigned char *wimg, int n0, int wave, int lane) {
    using C = RsCfg<K, DGRAD>;
    constexpr int CH = C::W_RB / 16, RPI = 64 / CH;          // chunks per row, rows per wave-instruction
#pragma unroll
    for (int i = 0; i < C::W_DMA; ++i) {
        const int inst = wave * C::W_DMA + i;
        const int r = inst * RPI + lane / CH, s = lane % CH;
        if (!DGRAD) {       // row = output column n0 + r, chunks along k
            const int c = s ^ (r & 15);
            const int gr = min(n0 + r, a.N - 1);
            dma16(a.W + (int64_t)gr * a.ldw + c * 8, wimg + inst * 1024);
        } else {            // row = reduction index r, chunks along the block's 128 output columns
            const int c = s ^ (2 * (r & 7));
            dma16(a.W + (int64_t)r * a.ldw + n0 + c * 8, wimg + inst * 1024);
        }
    }
}

template <int K, bool DGRAD>
__device__ __forceinline__ void issue_a(const RsArgs &a, unsigned char *slot, int m0, int wave, int lane) {
    using C = RsCfg<K, DGRAD>;
    constexpr int CH = C::RB / 16, RPI = 64 / CH;
#pragma unroll
    for (int i = 0; i < C::A_DMA; ++i) {
        const int inst = wave * C::A_DMA + i;
        const int r = inst * RPI + lane / CH, s = lane % CH;
        const int c = s ^ (r & 15);
        const int gr = min(m0 + r, a.M - 1);
        dma16(a.A + (int64_t)gr * a.lda + c * 8, slot + inst * 1024);
    }
}

template <int K, bool DGRAD, bool LN>
__global__ __launch_bounds__(256) void rowstream_kernel(const RsArgs a) {
    using C = RsCfg<K, DGRAD>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TM = 2, TN = 4, KS = K / 32;               // wave tile 32 x 64 (waves 2 x 2)
    unsigned char *Wimg = smem, *ring = smem + C::W_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, cq = lane & 15, q = cq >> 2, p = cq & 3;
    const int n0 = blockIdx.x * BN;
    const int ntm = (a.M + BM - 1) / BM;
    int mt = blockIdx.y;
    if (mt >= ntm) return;

    issue_w<K, DGRAD>(a, Wimg, n0, wave, lane);
    issue_a<K, DGRAD>(a, ring, mt * BM, wave, lane);
    const bool vec_ok = (a.ldc % 8 == 0) && (!a.resid || a.ldr % 8 == 0) && (!a.dact || a.ldd % 8 == 0) &&
                        (!LN || a.ldy % 8 == 0);

    for (int t = 0; mt < ntm; mt += gridDim.y, ++t) {
        const int m0 = mt * BM;
        wait_all_vm();          // this wave's share of tile t (and of W) has landed
        barrier();              // ... everyone's; and everyone is done with the staging area of tile t-1
        if (mt + (int)gridDim.y < ntm) issue_a<K, DGRAD>(a, ring + ((t + 1) & 1) * C::A_BYTES, (mt + gridDim.y) * BM, wave, lane);
        const unsigned char *As = ring + (t & 1) * C::A_BYTES;

        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * 32 + i * 16 + cq;
                const int c = (ks * 4 + g) ^ (r & 15);
                af[i] = *reinterpret_cast<const bf16x8 *>(As + r * C::RB + c * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (!DGRAD) {
                    const int n = wn * 64 + j * 16 + cq;
                    const int c = (ks * 4 + g) ^ (n & 15);
                    bf[j] = *reinterpret_cast<const bf16x8 *>(Wimg + n * C::W_RB + c * 16);
                } else {
                    // B[k = red][n]: k slots 0..3 = rows ks*32 + 8g + 0..3, slots 4..7 = + 4..7; this lane supplies
                    // columns ncol .. ncol + 3 of row (+ q)
                    const int ncol = wn * 64 + j * 16 + 4 * p;
                    s16x4 lo, hi;
                    {
                        const int r = ks * 32 + 8 * g + q;
                        const int c = (ncol >> 3) ^ (2 * (r & 7));
                        lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4 *)(Wimg + r * C::W_RB + c * 16 + (ncol & 7) * 2));
                    }
                    {
                        const int r = ks * 32 + 8 * g + 4 + q;
                        const int c = (ncol >> 3) ^ (2 * (r & 7));
                        hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4 *)(Wimg + r * C::W_RB + c * 16 + (ncol & 7) * 2));
                    }
                    struct { s16x4 x, y; } pr = {lo, hi};
                    bf[j] = __builtin_bit_cast(bf16x8, pr);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mma(af[i], bf[j], acc[i][j]);
        }
        // ---- epilogue: 32 rows per pass through fp32 staging in the ring slot just consumed ----
        float *Cs = reinterpret_cast<float *>(ring + (t & 1) * C::A_BYTES);
        for (int h = 0; h < 2; ++h) {
            barrier();          // all waves finished reading the A slot (h = 0) / the previous pass (h = 1)
            if (wm == h) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int lc = wn * 64 + j * 16 + cq;
                        const float bv = (a.bias && n0 + lc < a.N) ? a.bias[n0 + lc] : 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) Cs[(i * 16 + 4 * g + r) * BN + lc] = acc[i][j][r] + bv;
                    }
            }
            barrier();
#pragma unroll
            for (int it = 0; it < 32 * (BN / 8) / NT; ++it) {        // 2 chunks of 8 columns per thread
                const int c = tid + it * NT;
                const int lr = c / (BN / 8), lc = (c % (BN / 8)) * 8;
                const int row = m0 + h * 32 + lr, col = n0 + lc;
                const bool live = row < a.M && col < a.N;
                float v[8];
                *reinterpret_cast<float4 *>(v) = *reinterpret_cast<const float4 *>(Cs + lr * BN + lc);
                *reinterpret_cast<float4 *>(v + 4) = *reinterpret_cast<const float4 *>(Cs + lr * BN + lc + 4);
                const bool fast = live && vec_ok && col + 8 <= a.N;
                if (fast) {
                    if (a.C2) store16(a.C2 + (int64_t)row * a.ldc + col, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = act_fwd(a.act, v[e]);
                    if (a.dact) {
                        float d[8];
                        load16(a.dact + (int64_t)row * a.ldd + col, d);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= act_bwd(a.dact_kind, d[e]);
                    }
                    if (a.resid) {
                        float d[8];
                        load16(a.resid + (int64_t)row * a.ldr + col, d);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += d[e];
                    }
                    store16(a.C + (int64_t)row * a.ldc + col, v);
                } else if (live) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (col + e < a.N) {
                            float x = v[e];
                            if (a.C2) a.C2[(int64_t)row * a.ldc + col + e] = from_f32<bf16>(x);
                            x = act_fwd(a.act, x);
                            if (a.dact) x *= act_bwd(a.dact_kind, to_f32(a.dact[(int64_t)row * a.ldd + col + e]));
                            if (a.resid) x += to_f32(a.resid[(int64_t)row * a.ldr + col + e]);
                            v[e] = x;
                            a.C[(int64_t)row * a.ldc + col + e] = from_f32<bf16>(x);
                        }
                    }
                }
                if (LN) {
                    // the 16 threads holding one row are consecutive lanes: statistics by 4 shuffles.  LayerNorm acts on
                    // the values as STORED (bf16-rounded z), exactly like the unfused LayerNorm kernel reading z back.
                    float s1 = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] = live ? to_f32(from_f32<bf16>(v[e])) : 0.f;
                        s1 += v[e];
                    }
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
                    const float mu = s1 * (1.f / BN);
                    float s2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float dlt = v[e] - mu; s2 += dlt * dlt; }
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
                    const float rs = rsqrtf(s2 * (1.f / BN) + 1e-5f);
                    if (live) {
                        float o8[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) o8[e] = (v[e] - mu) * rs * a.gamma[lc + e] + a.beta[lc + e];
                        store16(a.Y + (int64_t)row * a.ldy + lc, o8);
                        if ((tid & 15) == 0) { a.mean[row] = mu; a.rstd[row] = rs; }
                    }
                }
            }
        }
    }
}

template <int K, bool DGRAD, bool LN>
int rs_launch(const RsArgs &a, hipStream_t s) {
    using C = RsCfg<K, DGRAD>;
    auto kern = rowstream_kernel<K, DGRAD, LN>;
    MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    const int ntn = ceil_div(a.N, BN), ntm = ceil_div(a.M, BM);
    const int per_cu = C::LDS <= 80 * 1024 ? 2 : 1;          // resident blocks per CU by LDS
    int gy = (256 * per_cu + ntn - 1) / ntn;
    if (gy > ntm) gy = ntm;
    ProfScope prof(s);
    hipLaunchKernelGGL(kern, dim3(ntn, gy), dim3(NT), C::LDS, s, a);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

bool rowstream_supported(int M, int N, int K, bool dgrad, int64_t lda, int64_t ldw, const void *A, const void *W) {
    static const bool off = getenv("MIVIT_NO_ROWSTREAM") != nullptr;
    if (off) return false;
    if (!(K == 128 || K == 256) || N % 128 != 0 || M < 256) return false;
    if (lda % 8 || ldw % 8 || !aligned16(A) || !aligned16(W)) return false;
    (void)dgrad;
    return true;
}

// forward / dgrad in one entry: C = epilogue(A . W)
int launch_rowstream(bool dgrad, const void *A, int64_t lda, const void *W_bf16, int64_t ldw, int M, int N, int K,
                     const float *bias, int act, const void *dact, int64_t ldd, int dact_kind, const void *resid,
                     int64_t ldr, void *Cout, int64_t ldc, void *C2, const float *gamma, const float *beta, void *Y,
                     int64_t ldy, float *mean, float *rstd, hipStream_t s) {
    RsArgs a = {static_cast<const bf16 *>(A), lda, static_cast<const bf16 *>(W_bf16), ldw, M, N, K, bias, act,
                static_cast<const bf16 *>(dact), ldd, dact_kind, static_cast<const bf16 *>(resid), ldr,
                static_cast<bf16 *>(Cout), ldc, static_cast<bf16 *>(C2), gamma, beta, static_cast<bf16 *>(Y), ldy, mean, rstd};
    const bool ln = gamma != nullptr;
    MIVIT_CHECK(!ln || (N == 128 && !dgrad), "rowstream: fused LayerNorm needs N == 128 (forward)");
    if (K == 128) {
        if (dgrad) return rs_launch<128, true, false>(a, s);
        return ln ? rs_launch<128, false, true>(a, s) : rs_launch<128, false, false>(a, s);
    }
    if (dgrad) return rs_launch<256, true, false>(a, s);
    return ln ? rs_launch<256, false, true>(a, s) : rs_launch<256, false, false>(a, s);
}
