// bf16 "wave-stream" GEMMs of the encoder-layer projections (K = 64 / 128 / 256 contraction -- 192 for the q|k|v data gradient
// of 64-wide models --, 128- or 64-column slices):
//      forward : C = act(A W^T + b) (+ residual) (+ fused post-norm LayerNorm when the slice is the whole row)
//      dgrad   : C = (A W) * act'(saved)  |  + residual gradient
// Same arithmetic and interface as the row-stream kernels (rowstream.hip), different data movement: these launches are
// HBM-bound on the activation rows, so the goal is bytes in flight with nothing to stall on.
//   * the 128-column weight slice is copied into LDS ONCE per workgroup (the only __syncthreads of the kernel);
//   * every wave then walks its own 16-row tiles: the A fragments go straight from global memory into the MFMA operand
//     registers (16 bytes per lane, one contiguous 64-byte run per row and instruction), prefetched 8 loads deep;
//   * the epilogue is wave-private (fp32 scratch of 16 x 128 per wave, row statistics of the LayerNorm by 2 shuffles),
//     so waves never wait for each other: no ring, no barrier, no DMA bookkeeping on the streaming path;
//   * dgrad reads the natural [n][k] weight image with ds_read_b64_tr_b16.
//
// AF32 (round 3): the forward kernel as the LINEAR FRAME EMBEDDING of small frames (reference helpers/models.py:146-164; patch
// sizes up to 16 x 16: 9 x 9 = 81 and 13 x 13 = 169 pixels in the shipped configurations; wider frames use embed.hip): A is the
// fp32 frame matrix [M, Kt] with ANY Kt <= K -- rows are only 4-byte aligned --, read through a raw buffer (16-byte loads at dword
// alignment; the range check returns zero past the end of the tensor), converted to the 16-bit element type in registers with
// the pixels past Kt (the next row's) zeroed; the weight slice [BN, Kt] is copied element-wise into a zero-padded image.  These
// shapes ran on the general register-staged GEMM with fp32 A before: 65 us of the 1.55 ms Framerate-shape step.
#include "common.h"
#include "stream_prims.h"
#include <stdlib.h>
#include "elem.h"          // element type of this translation unit (bf16, or f16 under -DMIVIT_ELEM_F16): after every other include

namespace {



struct WsArgs {
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
    int Kt;                        // AF32 only: the true row length of the fp32 A (= ldw of W), <= K
};


template <int K, int BN, bool DGRAD>
struct WsCfg {
    static constexpr int NWV = K == 64 ? 16 : (K == 128 ? 12 : 8);     // waves per workgroup (one workgroup per CU)
    static constexpr int PF = K == 64 ? 4 : (K == 128 ? 2 : 1);         // tiles of A fragments in flight per wave (8 loads)
    static constexpr int KS = K / 32;
    static constexpr int LDC = BN + 4;
    static constexpr int W_LD = DGRAD ? BN + 16 : K + 8;                // elements per LDS row of the weight image
    static constexpr int W_ROWS = DGRAD ? K : BN;
    static constexpr int W_BYTES = W_ROWS * W_LD * 2;
    static constexpr int SCRATCH = 16 * LDC * 4;
    static constexpr int GB_BYTES = 2 * BN * 4;                          // LayerNorm gamma | beta of the slice (fused-LayerNorm launches)
    static constexpr int LDS = W_BYTES + NWV * SCRATCH + GB_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS budget");
};


// E_KIND: 0 none, 1 residual add (forward), 2 activation-derivative multiply (dgrad), 3 residual add of a gradient
typedef unsigned int ws_u32x4 __attribute__((ext_vector_type(4)));
template <int K, int BN, bool DGRAD, bool LN, int E_KIND, bool AF32 = false>
__global__ __launch_bounds__((K == 64 ? 16 : (K == 128 ? 12 : 8)) * 64) void wavestream_kernel(const WsArgs a) {
    using C = WsCfg<K, BN, DGRAD>;
    constexpr int LDC = C::LDC, TN = BN / 16, NCH = BN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KS = C::KS, PF = C::PF, NT = C::NWV * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cq = lane & 15, q = cq >> 2, p = cq & 3;
    // XCD-aware placement (see rowstream.hip): the column slices of one row walker on the same XCD
    int bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.x > 1) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, G = 8 * gridDim.x;
        if (lin < (int)(gridDim.x * gridDim.y) / G * G) { bx = (lin % G) / 8; by = (lin / G) * 8 + lin % 8; }
    }
    const int n0 = bx * BN;
    bf16 *Wimg = reinterpret_cast<bf16 *>(smem);
    float *Cs = reinterpret_cast<float *>(smem + C::W_BYTES) + wave * 16 * LDC;
    float *gb = reinterpret_cast<float *>(smem + C::W_BYTES + C::NWV * C::SCRATCH);          // [gamma BN | beta BN]
    if (LN) {        // staged once: read from global memory in the epilogue they were 2 x 8 x NCH loads per 16-row tile and lane
        for (int i = tid; i < BN; i += NT) { gb[i] = a.gamma[i]; gb[BN + i] = a.beta[i]; }
    }

    // ---- weight slice -> LDS, once ----
    if (AF32) {         // W [N, Kt], rows 2-byte aligned: element-wise into the zero-padded image
        for (int i = tid; i < BN * K; i += NT) {
            const int n = i / K, k = i - n * K;
            bf16 w; w.v = 0;
            if (k < a.Kt) w = a.W[(int64_t)(n0 + n) * a.ldw + k];
            Wimg[n * C::W_LD + k] = w;
        }
    } else if (!DGRAD) {       // rows n0 .. n0+127 of W [N, K]: k-contiguous
        for (int i = tid; i < BN * (K / 8); i += NT) {
            const int n = i / (K / 8), c = i - n * (K / 8);
            *reinterpret_cast<uint4 *>(Wimg + n * C::W_LD + c * 8) = *reinterpret_cast<const uint4 *>(a.W + (int64_t)(n0 + n) * a.ldw + c * 8);
        }
    } else {            // rows 0 .. K-1 (contraction index), columns n0 .. n0+127 of W [K, N_total]
        for (int i = tid; i < K * (BN / 8); i += NT) {
            const int r = i / (BN / 8), c = i - r * (BN / 8);
            *reinterpret_cast<uint4 *>(Wimg + r * C::W_LD + c * 8) = *reinterpret_cast<const uint4 *>(a.W + (int64_t)r * a.ldw + n0 + c * 8);
        }
    }
    __syncthreads();

    // this lane's bias values, loaded once per wave (read inside the tile loop under `a.bias ? ... : 0` they were TN serialised
    // L2 round trips per 16-row tile, each behind a wait that also drained the row prefetch)
    float bias_v[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bias_v[j] = 0.f;
    if (a.bias) {
#pragma unroll
        for (int j = 0; j < TN; ++j) bias_v[j] = a.bias[n0 + j * 16 + cq];
    }
    const int ntiles = (a.M + 15) / 16, stride = gridDim.y * C::NWV;
    int tile = by * C::NWV + wave;
    const bf16 *E = E_KIND == 2 ? a.dact : a.resid;
    const int64_t lde = E_KIND == 2 ? a.ldd : a.ldr;

    bf16x8 nx[PF][AF32 ? 2 * KS : KS];          // (AF32: the raw fp32 pixels, two 16-byte halves per fragment)
    // AF32: the frames as one raw buffer (the launcher checks M * Kt * 4 < 2^32)
    const __amdgpu_buffer_rsrc_t arows = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(a.A), 0, AF32 ? (int)((unsigned)a.M * (unsigned)a.Kt * 4u) : 0, 0x00020000);
    auto load_tile = [&](int t, bf16x8 (&dst)[AF32 ? 2 * KS : KS]) {
        if constexpr (AF32) {
            const int off = (int)(((unsigned)min(t * 16 + cq, a.M - 1) * (unsigned)a.Kt + 8u * g) * 4u);          // (< 2^32: unsigned arithmetic)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                dst[2 * ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(arows, off + ks * 128, 0, 0));
                dst[2 * ks + 1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(arows, off + ks * 128 + 16, 0, 0));
            }
        } else {
            const bf16 *ap = a.A + (int64_t)min(t * 16 + cq, a.M - 1) * a.lda + 8 * g;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dst[ks] = *reinterpret_cast<const bf16x8 *>(ap + ks * 32);
        }
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) load_tile(min(tile + i * stride, ntiles - 1), nx[i]);

    for (; tile < ntiles; tile += stride) {
        bf16x8 cur[KS];
        if constexpr (AF32) {          // fp32 pixels -> element type; pixels past Kt (the next row's) are zeroed
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const ws_u32x4 lo = __builtin_bit_cast(ws_u32x4, nx[0][2 * ks]), hi = __builtin_bit_cast(ws_u32x4, nx[0][2 * ks + 1]);
                const uint32_t w[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) cur[ks][e] = (__bf16)(ks * 32 + 8 * g + e < a.Kt ? __uint_as_float(w[e]) : 0.f);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) cur[ks] = nx[0][ks];
        }
#pragma unroll
        for (int i = 0; i + 1 < PF; ++i)
#pragma unroll
            for (int ks = 0; ks < (AF32 ? 2 * KS : KS); ++ks) nx[i][ks] = nx[i + 1][ks];
        load_tile(min(tile + PF * stride, ntiles - 1), nx[PF - 1]);

        f32x4 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bf16x8 bf;
                if (!DGRAD) bf = *reinterpret_cast<const bf16x8 *>(Wimg + (j * 16 + cq) * C::W_LD + ks * 32 + 8 * g);
                else bf = tr_pair(Wimg + (ks * 32 + 8 * g + q) * C::W_LD + j * 16 + 4 * p,
                                  Wimg + (ks * 32 + 8 * g + 4 + q) * C::W_LD + j * 16 + 4 * p);
                acc[j] = mma(cur[ks], bf, acc[j]);
            }
        }
        // ---- wave-private epilogue: accumulators (+ bias) -> fp32 scratch -> rows of BN, four lanes per row ----
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int lc = j * 16 + cq;
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(4 * g + r) * LDC + lc] = acc[j][r] + bias_v[j];
        }
        wave_lds_fence();
        const int lr = lane >> 2, row = tile * 16 + lr;
        const bool live = row < a.M;
        float v[NCH][8];
        float s1 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int lc = c * 32 + (lane & 3) * 8, col = n0 + lc;
            *reinterpret_cast<float4 *>(v[c]) = *reinterpret_cast<const float4 *>(Cs + lr * LDC + lc);
            *reinterpret_cast<float4 *>(v[c] + 4) = *reinterpret_cast<const float4 *>(Cs + lr * LDC + lc + 4);
            if (live && a.C2) store16(a.C2 + (int64_t)row * a.ldc + col, v[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = act_fwd(a.act, v[c][e]);
            if (E_KIND != 0) {
                float d[8];
                load16(E + (int64_t)(live ? row : 0) * lde + col, d);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c][e] = E_KIND == 2 ? v[c][e] * act_bwd(a.dact_kind, d[e]) : v[c][e] + d[e];
            }
            if (live) store16(a.C + (int64_t)row * a.ldc + col, v[c]);
            if (LN) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {       // LayerNorm acts on the values as STORED (bf16-rounded z)
                    v[c][e] = live ? to_f32(from_f32<bf16>(v[c][e])) : 0.f;
                    s1 += v[c][e];
                }
            }
        }
        if (LN) {
            s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
            const float mu = s1 * (1.f / BN);
            float s2 = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float dlt = v[c][e] - mu; s2 += dlt * dlt; }
            s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
            const float rs = rsqrtf(s2 * (1.f / BN) + 1e-5f);
            if (live) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int lc = c * 32 + (lane & 3) * 8;
                    float o8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o8[e] = (v[c][e] - mu) * rs * gb[lc + e] + gb[BN + lc + e];
                    store16(a.Y + (int64_t)row * a.ldy + lc, o8);
                }
                if ((lane & 3) == 0) { a.mean[row] = mu; a.rstd[row] = rs; }
            }
        }
    }
}

template <int K, int BN, bool DGRAD, bool LN, int E_KIND, bool AF32 = false>
int ws_launch(const WsArgs &a, hipStream_t s) {
    using C = WsCfg<K, BN, DGRAD>;
    auto kern = wavestream_kernel<K, BN, DGRAD, LN, E_KIND, AF32>;
    MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    const int ntn = a.N / BN, ntiles = ceil_div(a.M, 16);
    int gy = 256 / ntn;                                       // one resident workgroup per CU, no tail round
    if (gy < 1) gy = 1;
    if (gy > ceil_div(ntiles, C::NWV)) gy = ceil_div(ntiles, C::NWV);
    ProfScope prof(s);
    hipLaunchKernelGGL(kern, dim3(ntn, gy), dim3(C::NWV * 64), C::LDS, s, a);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

// epilogue variants of one (K, BN) shape
template <int K, int BN>
int ws_dispatch(const WsArgs &a, bool dgrad, bool ln, hipStream_t s) {
    if (dgrad) {
        if (a.dact) return ws_launch<K, BN, true, false, 2>(a, s);
        if (a.resid) return ws_launch<K, BN, true, false, 3>(a, s);
        return ws_launch<K, BN, true, false, 0>(a, s);
    }
    if (ln) return ws_launch<K, BN, false, true, 1>(a, s);
    if (a.resid) return ws_launch<K, BN, false, false, 1>(a, s);
    return ws_launch<K, BN, false, false, 0>(a, s);
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

bool wavestream_supported(int M, int N, int K, bool dgrad, int64_t lda, int64_t ldw, const void *A, const void *W) {
    static const bool off = getenv("MIVIT_NO_WAVESTREAM") != nullptr;
    if (off) return false;
    // (K = 192: the q|k|v data gradient of the 64-wide models -- contraction over 3 E = 192, 64 output columns; it ran on the
    //  general register-staged GEMM before: 54 us per layer at the Framerate shape against ~20 us for its neighbours)
    if (!(K == 64 || K == 128 || K == 256 || (K == 192 && dgrad && N == 64)) || N % 64 != 0 || M < 256) return false;
    if (K == 256 && N % 128 != 0) return false;
    if (lda % 8 || ldw % 8 || !aligned16(A) || !aligned16(W)) return false;
    return true;
}

// same contract as launch_rowstream (rowstream.hip)
int launch_wavestream(bool dgrad, const void *A, int64_t lda, const void *W_bf16, int64_t ldw, int M, int N, int K,
                      const float *bias, int act, const void *dact, int64_t ldd, int dact_kind, const void *resid,
                      int64_t ldr, void *Cout, int64_t ldc, void *C2, const float *gamma, const float *beta, void *Y,
                      int64_t ldy, float *mean, float *rstd, hipStream_t s) {
    WsArgs a = {static_cast<const bf16 *>(A), lda, static_cast<const bf16 *>(W_bf16), ldw, M, N, K, bias, act,
                static_cast<const bf16 *>(dact), ldd, dact_kind, static_cast<const bf16 *>(resid), ldr,
                static_cast<bf16 *>(Cout), ldc, static_cast<bf16 *>(C2), gamma, beta, static_cast<bf16 *>(Y), ldy, mean, rstd};
    const bool ln = gamma != nullptr;
    const int BNsel = N % 128 == 0 ? 128 : 64;
    MIVIT_CHECK(!ln || (N == BNsel && !dgrad && resid), "wavestream: fused LayerNorm needs the slice to be the whole row (N = 64 or 128), forward, with a residual");
    MIVIT_CHECK(!(dact && resid), "wavestream: at most one epilogue operand");
    MIVIT_CHECK(ldc % 8 == 0 && (!ln || ldy % 8 == 0) && aligned16(Cout) && (!C2 || aligned16(C2)) && (!Y || aligned16(Y)),
                "wavestream: outputs must be 16-byte aligned with ld % 8 == 0");
    MIVIT_CHECK((!resid || (ldr % 8 == 0 && aligned16(resid))) && (!dact || (ldd % 8 == 0 && aligned16(dact))),
                "wavestream: epilogue operand must be 16-byte aligned with ld % 8 == 0");
    if (BNsel == 128) {
        if (K == 64) return ws_dispatch<64, 128>(a, dgrad, ln, s);
        if (K == 128) return ws_dispatch<128, 128>(a, dgrad, ln, s);
        return ws_dispatch<256, 128>(a, dgrad, ln, s);
    }
    if (K == 64) return ws_dispatch<64, 64>(a, dgrad, ln, s);
    if (K == 192) {
        MIVIT_CHECK(dgrad, "wavestream: K = 192 is a data-gradient shape");
        if (a.dact) return ws_launch<192, 64, true, false, 2>(a, s);
        if (a.resid) return ws_launch<192, 64, true, false, 3>(a, s);
        return ws_launch<192, 64, true, false, 0>(a, s);
    }
    return ws_dispatch<128, 64>(a, dgrad, ln, s);
}

// ---- linear frame embedding of small frames: Y [M, E] = X [M, K] (fp32, any K <= 256) W^T + b ----
static int frame_kp(int K) { return K <= 64 ? 64 : K <= 96 ? 96 : K <= 128 ? 128 : K <= 192 ? 192 : K <= 256 ? 256 : 0; }
bool embed_small_fwd_supported(int M, int K, int E, const void *X, const void *Y) {
    static const bool off = getenv("MIVIT_NO_EMBED_SMALL") != nullptr;
    return !off && frame_kp(K) != 0 && (E == 64 || E == 128) && M >= 256 && (int64_t)M * K * 4 < (1ll << 32) &&
           (reinterpret_cast<uintptr_t>(X) & 3) == 0 && aligned16(Y);
}
int launch_embed_small_fwd(const float *X, const void *W_bf16, const float *bias, void *Y, int M, int K, int E, hipStream_t s) {
    WsArgs a = {};
    a.A = reinterpret_cast<const bf16 *>(X); a.lda = K; a.W = static_cast<const bf16 *>(W_bf16); a.ldw = K;
    a.M = M; a.N = E; a.K = frame_kp(K); a.Kt = K; a.bias = bias; a.act = MIVIT_ACT_NONE;
    a.C = static_cast<bf16 *>(Y); a.ldc = E;
#define EMB_SMALL(KP_)                                                                            \
    do {                                                                                          \
        if (E == 64) return ws_launch<KP_, 64, false, false, 0, true>(a, s);                      \
        return ws_launch<KP_, 128, false, false, 0, true>(a, s);                                  \
    } while (0)
    switch (a.K) {
        case 64: EMB_SMALL(64);
        case 96: EMB_SMALL(96);
        case 128: EMB_SMALL(128);
        case 192: EMB_SMALL(192);
        case 256: EMB_SMALL(256);
    }
#undef EMB_SMALL
    MIVIT_FAIL("embed_small_fwd: unsupported frame size %d", K);
}

#ifndef MIVIT_ELEM_F16      // operator-level C-ABI: declared for bf16 (include/mivit_hip.h)
extern "C" int mivit_embed_small_supported(int M, int K, int E) {
    return frame_kp(K) != 0 && (E == 64 || E == 128) && M >= 256 && (int64_t)M * K * 4 < (1ll << 32) && !getenv("MIVIT_NO_EMBED_SMALL");
}
extern "C" int mivit_embed_small_fwd(const float *X, const void *W_bf16, const float *bias, int M, int K, int E, void *Y, void *stream) {
    MIVIT_CHECK(X && W_bf16 && Y, "embed_small_fwd: null pointer");
    if (!embed_small_fwd_supported(M, K, E, X, Y)) { mivit_set_error("embed_small_fwd: unsupported shape"); return 3; }
    prof_set_tag(MIVIT_PROF_OP);
    return launch_embed_small_fwd(X, W_bf16, bias, Y, M, K, E, static_cast<hipStream_t>(stream));
}
extern "C" int mivit_wavestream_fwd(const void *x, int64_t ldx, const void *W_bf16, const float *bias, int M, int N, int K,
                                    int act, const void *resid, int64_t ldr, void *y, int64_t ldy, void *y_preact,
                                    const float *ln_gamma, const float *ln_beta, void *ln_out, float *mean, float *rstd,
                                    void *stream) {
    MIVIT_CHECK(x && W_bf16 && y, "wavestream_fwd: null pointer");
    if (!wavestream_supported(M, N, K, false, ldx, K, x, W_bf16)) { mivit_set_error("wavestream_fwd: unsupported shape"); return 3; }
    prof_set_tag(MIVIT_PROF_OP);
    return launch_wavestream(false, x, ldx, W_bf16, K, M, N, K, bias, act, nullptr, 0, 0, resid, ldr, y, ldy, y_preact, ln_gamma,
                             ln_beta, ln_out, N, mean, rstd, static_cast<hipStream_t>(stream));
}
extern "C" int mivit_wavestream_dgrad(const void *dy, int64_t lddy, const void *W_bf16, int M, int N, int K, int act,
                                      const void *saved, int64_t lds, const void *dres, int64_t lddr, void *dx, int64_t lddx,
                                      void *stream) {
    MIVIT_CHECK(dy && W_bf16 && dx, "wavestream_dgrad: null pointer");
    if (!wavestream_supported(M, K, N, true, lddy, K, dy, W_bf16)) { mivit_set_error("wavestream_dgrad: unsupported shape"); return 3; }
    prof_set_tag(MIVIT_PROF_OP);
    return launch_wavestream(true, dy, lddy, W_bf16, K, M, K, N, nullptr, MIVIT_ACT_NONE, act != MIVIT_ACT_NONE ? saved : nullptr,
                             lds, act, dres, lddr, dx, lddx, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                             static_cast<hipStream_t>(stream));
}
#endif
