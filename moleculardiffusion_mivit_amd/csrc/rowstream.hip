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
#include "stream_prims.h"
#include <stdlib.h>
#include "elem.h"          // element type of this translation unit (bf16, or f16 under -DMIVIT_ELEM_F16): after every other include

namespace {



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
    int xcd_remap;
};

constexpr int BN = 128, NW = 8, NT = NW * 64;

template <int K, bool DGRAD, bool HAS_E>
struct RsCfg {
    static constexpr int BM = K > 256 ? 32 : 64;             // rows per tile (the K = 384 weight slice leaves less LDS)
    static constexpr int E_BYTES = BM * BN * 2;              // optional epilogue-operand tile (residual or saved act.)
    static constexpr int WMW = BM / 32, WNW = NW / WMW;      // wave grid: WMW x WNW, wave tile 32 x (128 / WNW)
    static constexpr int TN = BN / WNW / 16;
    static constexpr int RB = 2 * K;                         // bytes per A row (and per W row in forward)
    static constexpr int A_BYTES = BM * RB;
    static constexpr int SLOT = A_BYTES + (HAS_E ? E_BYTES : 0);
    static constexpr int W_ROWS = DGRAD ? K : BN;
    static constexpr int W_RB = DGRAD ? BN * 2 : RB;
    static constexpr int W_BYTES = W_ROWS * W_RB;
    static constexpr int NS = (W_BYTES + 3 * SLOT <= 160 * 1024) ? 3 : 2;     // ring slots (NS - 1 tiles in flight)
    static constexpr int LDS = W_BYTES + NS * SLOT;
    static constexpr int A_DMA = A_BYTES / 1024 / NW;        // wave-instructions per wave per tile
    static constexpr int E_DMA = HAS_E ? E_BYTES / 1024 / NW : 0;
    static constexpr int W_DMA = W_BYTES / 1024 / NW;
    static constexpr int PER_TILE = A_DMA + E_DMA;
    static_assert(A_BYTES >= 32 * BN * 4, "fp32 staging of 32 rows must fit in the A part of a ring slot");
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int K, bool DGRAD, bool HAS_E>
__device__ __forceinline__ void issue_w(const RsArgs &a, unsigned char *wimg, int n0, int wave, int lane) {
    using C = RsCfg<K, DGRAD, HAS_E>;
    constexpr int CH = C::W_RB / 16;                         // 16-byte chunks per row; the image is chunk-linear
#pragma unroll
    for (int i = 0; i < C::W_DMA; ++i) {
        const int inst = wave * C::W_DMA + i;
        const int ci = inst * 64 + lane, r = ci / CH, s = ci % CH;
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

// one tile = 64 rows of A (swizzled) [+ the 64 x 128 epilogue-operand tile, linear]
template <int K, bool DGRAD, bool HAS_E>
__device__ __forceinline__ void issue_tile(const RsArgs &a, const bf16 *E, int64_t lde, unsigned char *slot, int m0, int n0,
                                           int wave, int lane) {
    using C = RsCfg<K, DGRAD, HAS_E>;
    constexpr int CH = C::RB / 16;
#pragma unroll
    for (int i = 0; i < C::A_DMA; ++i) {
        const int inst = wave * C::A_DMA + i;
        const int ci = inst * 64 + lane, r = ci / CH, s = ci % CH;
        const int c = s ^ (r & 15);
        const int gr = min(m0 + r, a.M - 1);
        dma16(a.A + (int64_t)gr * a.lda + c * 8, slot + inst * 1024);
    }
    if (HAS_E) {
        unsigned char *es = slot + C::A_BYTES;
#pragma unroll
        for (int i = 0; i < C::E_DMA; ++i) {
            const int inst = wave * C::E_DMA + i;
            const int r = inst * 4 + (lane >> 4), s = lane & 15;
            const int gr = min(m0 + r, a.M - 1);
            dma16(E + (int64_t)gr * lde + n0 + s * 8, es + inst * 1024);
        }
    }
}

// E_KIND: 0 none, 1 residual add (forward), 2 activation-derivative multiply (dgrad), 3 residual add of a gradient
template <int K, bool DGRAD, bool LN, int E_KIND>
__global__ __launch_bounds__(NT) void rowstream_kernel(const RsArgs a) {
    constexpr bool HAS_E = E_KIND != 0;
    using C = RsCfg<K, DGRAD, HAS_E>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TM = 2, TN = C::TN, KS = K / 32, D = C::NS - 1;
    unsigned char *Wimg = smem, *ring = smem + C::W_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WNW, wn = wave % C::WNW;
    constexpr int WCOLS = BN / C::WNW;                       // columns per wave
    const int g = lane >> 4, cq = lane & 15, q = cq >> 2, p = cq & 3;
    constexpr int BM = C::BM;
    // XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs (each with its own L2) in dispatch order
    // (blockIdx.x fastest).  The column slices of one row walker re-read the same A rows, so they are placed on the SAME
    // XCD: within a group of 8 * gridDim.x consecutive workgroups, XCD k hosts all slices of walker 8 * group + k.
    int bx = blockIdx.x, by = blockIdx.y;
    if (a.xcd_remap && gridDim.x > 1) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, G = 8 * gridDim.x;
        if (lin < (int)(gridDim.x * gridDim.y) / G * G) { bx = (lin % G) / 8; by = (lin / G) * 8 + lin % 8; }
    }
    const int n0 = bx * BN;
    const int ntm = (a.M + BM - 1) / BM;
    const int stride = gridDim.y;
    int mt = by;
    if (mt >= ntm) return;
    const bf16 *E = E_KIND == 2 ? a.dact : a.resid;
    const int64_t lde = E_KIND == 2 ? a.ldd : a.ldr;

    // this lane's bias values, loaded once (inside the tile loop, under `a.bias ? ... : 0`, each was an L2 round trip behind a
    // vmcnt(0) -- which also drained the next tile's DMA prefetch in every epilogue pass)
    float bias_v[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bias_v[j] = 0.f;
    if (a.bias) {
#pragma unroll
        for (int j = 0; j < TN; ++j) bias_v[j] = a.bias[n0 + wn * WCOLS + j * 16 + cq];
    }
    issue_w<K, DGRAD, HAS_E>(a, Wimg, n0, wave, lane);
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (mt + s * stride < ntm) issue_tile<K, DGRAD, HAS_E>(a, E, lde, ring + s * C::SLOT, (mt + s * stride) * BM, n0, wave, lane);

    for (int t = 0; mt < ntm; mt += stride, ++t) {
        const int m0 = mt * BM;
        // Tile t must have landed.  VM program order of a wave (T = the PER_TILE DMA pieces of a tile, st = the row stores of one
        // epilogue: at least ST = BM / 32 store instructions, one unconditional-for-a-full-tile C store per 32-row pass; every
        // tile before the one being waited for is a full tile, so every wave issued them; W = the weight slice):
        //   prologue     W  T(0) .. T(D-1)
        //   iteration i  wait | barrier | T(i+D) if it exists | compute(i) | st(i)
        // D = 2:  W T0 T1 | T2 st0 | T3 st1 | ...   younger than T(t): T(t+1) [if it exists], st(t-2) [t >= 2], st(t-1) [t >= 1]
        // D = 1:  W T0 | T1 st0 | T2 st1 | ...      younger than T(t): st(t-1) [t >= 1]
        // (Round 2 counted "4 stores of the last epilogue" everywhere: with 32-row passes an epilogue guarantees BM / 32 = 1 or 2,
        // so at t = 1 (D = 2) and at every t (D = 1: the K = 384 slices, BM = 32, one store) the count let DMA pieces of the
        // tile being waited for stay in flight.)
        constexpr int ST = BM / 32;
        static_assert(D == 1 || D == 2, "the counts below are written out for one or two tiles in flight");
        const bool next_in_flight = D > 1 && mt + (D - 1) * stride < ntm;
        if (t == 0) { if (next_in_flight) wait_vm<(D - 1) * C::PER_TILE>(); else wait_vm<0>(); }
        else if (D == 1) wait_vm<ST>();
        else if (!next_in_flight) wait_vm<0>();
        else if (t == 1) wait_vm<C::PER_TILE + ST>();
        else wait_vm<C::PER_TILE + 2 * ST>();
        barrier();              // ... everyone's share landed; and everyone is done with the staging area of tile t-1
        if (mt + D * stride < ntm)
            issue_tile<K, DGRAD, HAS_E>(a, E, lde, ring + ((t + D) % C::NS) * C::SLOT, (mt + D * stride) * BM, n0, wave, lane);
        unsigned char *slot = ring + (t % C::NS) * C::SLOT;
        const unsigned char *As = slot, *Es = slot + C::A_BYTES;

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
                    const int n = wn * WCOLS + j * 16 + cq;
                    const int c = (ks * 4 + g) ^ (n & 15);
                    bf[j] = *reinterpret_cast<const bf16x8 *>(Wimg + n * C::W_RB + c * 16);
                } else {
                    // B[k = red][n]: k slots 0..3 = rows ks*32 + 8g + 0..3, slots 4..7 = + 4..7; this lane supplies
                    // columns ncol .. ncol + 3 of row (+ q)
                    const int ncol = wn * WCOLS + j * 16 + 4 * p;
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
        // ---- epilogue: 32 rows per pass through fp32 staging in the A part of the ring slot just consumed; every
        //      operand it needs is already in LDS, so it issues stores only ----
        float *Cs = reinterpret_cast<float *>(slot);
        for (int h = 0; h < BM / 32; ++h) {
            barrier();          // all waves finished reading the A image (h = 0) / the previous pass (h = 1)
            if (wm == h) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int lc = wn * WCOLS + j * 16 + cq;
#pragma unroll
                        for (int r = 0; r < 4; ++r) Cs[(i * 16 + 4 * g + r) * BN + lc] = acc[i][j][r] + bias_v[j];
                    }
            }
            barrier();
            {
                const int lr = tid / (BN / 8), lc = (tid % (BN / 8)) * 8;       // one 8-column chunk per thread
                const int row = m0 + h * 32 + lr, col = n0 + lc;
                const bool live = row < a.M;
                float v[8];
                *reinterpret_cast<float4 *>(v) = *reinterpret_cast<const float4 *>(Cs + lr * BN + lc);
                *reinterpret_cast<float4 *>(v + 4) = *reinterpret_cast<const float4 *>(Cs + lr * BN + lc + 4);
                if (live && a.C2) store16(a.C2 + (int64_t)row * a.ldc + col, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = act_fwd(a.act, v[e]);
                if (HAS_E) {
                    float d[8];
                    load16(reinterpret_cast<const bf16 *>(Es + (h * 32 + lr) * 256) + lc, d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = E_KIND == 2 ? v[e] * act_bwd(a.dact_kind, d[e]) : v[e] + d[e];
                }
                if (live) store16(a.C + (int64_t)row * a.ldc + col, v);
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

template <int K, bool DGRAD, bool LN, int E_KIND>
int rs_launch(const RsArgs &a, hipStream_t s) {
    using C = RsCfg<K, DGRAD, E_KIND != 0>;
    constexpr int BM = C::BM;
    auto kern = rowstream_kernel<K, DGRAD, LN, E_KIND>;
    MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    const int ntn = ceil_div(a.N, BN), ntm = ceil_div(a.M, BM);
    const int per_cu = C::LDS <= 80 * 1024 ? 2 : 1;          // resident blocks per CU by LDS
    int gy = (256 * per_cu) / ntn;                           // never more blocks than can be resident: no tail round
    if (gy < 1) gy = 1;
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
    if (!(K == 128 || K == 256 || (K == 384 && dgrad)) || N % 128 != 0 || M < 256) return false;
    if (lda % 8 || ldw % 8 || !aligned16(A) || !aligned16(W)) return false;
    return true;
}

// forward / dgrad in one entry: C = epilogue(A . W).  At most one of {resid, dact} (the engine never needs both).
int launch_rowstream(bool dgrad, const void *A, int64_t lda, const void *W_bf16, int64_t ldw, int M, int N, int K,
                     const float *bias, int act, const void *dact, int64_t ldd, int dact_kind, const void *resid,
                     int64_t ldr, void *Cout, int64_t ldc, void *C2, const float *gamma, const float *beta, void *Y,
                     int64_t ldy, float *mean, float *rstd, hipStream_t s) {
    // wave-stream variant (wavestream.hip): MIVIT_WAVESTREAM = 0 never, 1 whenever supported, default 2 = the launches where
    // it measured faster at the headline shape (out-proj + LayerNorm 81 -> 74 us, FC2 dgrad with act' 124 -> 93 us,
    // K = 256 dgrad 71 -> 68 us; the plain forward slices and FC2 + LayerNorm stay on the DMA ring: 61 vs 70, 88 vs 94 us)
    static const int use_ws = getenv("MIVIT_WAVESTREAM") ? atoi(getenv("MIVIT_WAVESTREAM")) : 2;
    const bool rs_shape = (K == 128 || K == 256 || (K == 384 && dgrad)) && N % 128 == 0;     // what the DMA-ring kernels cover
    static const int ws_mask = getenv("MIVIT_WAVESTREAM_MASK") ? atoi(getenv("MIVIT_WAVESTREAM_MASK")) : 7;      // (A/B: which of the three picks)
    const bool ws_pick = !rs_shape || use_ws == 1 || (use_ws == 2 && (((ws_mask & 1) && K == 128 && !dgrad && gamma) ||
                                                                      ((ws_mask & 2) && K == 128 && dgrad && dact) ||
                                                                      ((ws_mask & 4) && K == 256 && dgrad)));
    if (ws_pick && wavestream_supported(M, N, K, dgrad, lda, ldw, A, W_bf16))
        return launch_wavestream(dgrad, A, lda, W_bf16, ldw, M, N, K, bias, act, dact, ldd, dact_kind, resid, ldr, Cout, ldc, C2,
                                 gamma, beta, Y, ldy, mean, rstd, s);
    MIVIT_CHECK(rs_shape, "rowstream: unsupported shape M=%d N=%d K=%d", M, N, K);
    RsArgs a = {static_cast<const bf16 *>(A), lda, static_cast<const bf16 *>(W_bf16), ldw, M, N, K, bias, act,
                static_cast<const bf16 *>(dact), ldd, dact_kind, static_cast<const bf16 *>(resid), ldr,
                static_cast<bf16 *>(Cout), ldc, static_cast<bf16 *>(C2), gamma, beta, static_cast<bf16 *>(Y), ldy, mean, rstd, 0};
    static const int remap = getenv("MIVIT_XCD_REMAP") ? atoi(getenv("MIVIT_XCD_REMAP")) : 1;     // measured: forward slices 2.58 -> 2.50 ms/step
    a.xcd_remap = remap;
    const bool ln = gamma != nullptr;
    MIVIT_CHECK(!ln || (N == 128 && !dgrad && resid), "rowstream: fused LayerNorm needs N == 128, forward, with a residual");
    MIVIT_CHECK(!(dact && resid), "rowstream: at most one epilogue operand");
    MIVIT_CHECK(ldc % 8 == 0 && (!ln || ldy % 8 == 0) && aligned16(Cout) && (!C2 || aligned16(C2)) && (!Y || aligned16(Y)),
                "rowstream: outputs must be 16-byte aligned with ld % 8 == 0");
    MIVIT_CHECK((!resid || (ldr % 8 == 0 && aligned16(resid))) && (!dact || (ldd % 8 == 0 && aligned16(dact))),
                "rowstream: epilogue operand must be 16-byte aligned with ld % 8 == 0");
#define RS_GO(KK, DG, LNF, EK) return rs_launch<KK, DG, LNF, EK>(a, s)
    if (K == 128) {
        if (dgrad) { if (dact) RS_GO(128, true, false, 2); if (resid) RS_GO(128, true, false, 3); RS_GO(128, true, false, 0); }
        if (ln) RS_GO(128, false, true, 1);
        if (resid) RS_GO(128, false, false, 1);
        RS_GO(128, false, false, 0);
    }
    if (K == 384) { if (dact) RS_GO(384, true, false, 2); if (resid) RS_GO(384, true, false, 3); RS_GO(384, true, false, 0); }
    if (dgrad) { if (dact) RS_GO(256, true, false, 2); if (resid) RS_GO(256, true, false, 3); RS_GO(256, true, false, 0); }
    if (ln) RS_GO(256, false, true, 1);
    if (resid) RS_GO(256, false, false, 1);
    RS_GO(256, false, false, 0);
#undef RS_GO
}

#ifndef MIVIT_ELEM_F16      // operator-level C-ABI: declared for bf16 (include/mivit_hip.h)
extern "C" int mivit_rowstream_fwd(const void *x, int64_t ldx, const void *W_bf16, const float *bias, int M, int N, int K,
                                   int act, const void *resid, int64_t ldr, void *y, int64_t ldy, void *y_preact,
                                   const float *ln_gamma, const float *ln_beta, void *ln_out, float *mean, float *rstd,
                                   void *stream) {
    MIVIT_CHECK(x && W_bf16 && y, "rowstream_fwd: null pointer");
    if (!rowstream_supported(M, N, K, false, ldx, K, x, W_bf16)) { mivit_set_error("rowstream_fwd: unsupported shape"); return 3; }
    prof_set_tag(MIVIT_PROF_OP);
    return launch_rowstream(false, x, ldx, W_bf16, K, M, N, K, bias, act, nullptr, 0, 0, resid, ldr, y, ldy, y_preact, ln_gamma,
                            ln_beta, ln_out, N, mean, rstd, static_cast<hipStream_t>(stream));
}
extern "C" int mivit_rowstream_dgrad(const void *dy, int64_t lddy, const void *W_bf16, int M, int N, int K, int act,
                                     const void *saved, int64_t lds, const void *dres, int64_t lddr, void *dx, int64_t lddx,
                                     void *stream) {
    MIVIT_CHECK(dy && W_bf16 && dx, "rowstream_dgrad: null pointer");
    if (!rowstream_supported(M, K, N, true, lddy, K, dy, W_bf16)) { mivit_set_error("rowstream_dgrad: unsupported shape"); return 3; }
    prof_set_tag(MIVIT_PROF_OP);
    return launch_rowstream(true, dy, lddy, W_bf16, K, M, K, N, nullptr, MIVIT_ACT_NONE, act != MIVIT_ACT_NONE ? saved : nullptr,
                            lds, act, dres, lddr, dx, lddx, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                            static_cast<hipStream_t>(stream));
}
#endif
