// bf16 GEMMs of WIDE layers (embed_dim 512-class models: K, N beyond what the resident-weight row-stream kernels
// hold in LDS).  These shapes are MFMA/LDS-bound, not HBM-bound, so the design goal is MFMAs per LDS byte:
//   * workgroup tile (4 * BMW) x 128, four waves stacked along M, each wave a BMW x 128 sub-tile.  BMW = 64 has the better
//     MFMA : LDS ratio (12 fragment reads per 32 MFMAs = 96 B/clk at full MFMA rate, LDS limit 128) but needs 268
//     registers = one wave per SIMD, and nothing then hides the LDS-read -> MFMA dependency; BMW = 32 (141 registers,
//     three workgroups per CU) measures 1.5x faster and is the default (sweep: DESIGN.md);
//   * both operands global -> LDS by LDS-DMA (16 B per lane) into an NS-slot ring, NS - 1 stages in flight behind counted
//     s_waitcnt vmcnt + one raw s_barrier per stage; XOR swizzle of the 16-byte chunks on the SOURCE address;
//   * forward:  Y = act(A W^T + b) (+ residual), optional pre-activation copy      A [M,K], W [N,K]: both k-contiguous
//     dgrad  :  dX = (dY W) * act'(saved) (+ d residual)                           contraction over n: W tile natural
//               [n][k] image read with ds_read_b64_tr_b16
//   * epilogue through per-wave LDS scratch (fp32), 16-byte loads of residual / saved activation, 16-byte bf16 stores.
// Template parameters (BMW, BK, NS) are the "MFMA tile + LDS sizing" knobs of BASELINE config 4; the sweep is in
// scripts/sweep_gemm_dma.py and DESIGN.md.
#include "common.h"
#include "stream_prims.h"
#include <stdlib.h>
#include <type_traits>

namespace {



struct GdArgs {
    const bf16 *A; int64_t lda;        // [M, Kc]  (forward: x, Kc = K;  dgrad: dy, Kc = N_out)
    const bf16 *W; int64_t ldw;        // [N_out, K_in] bf16 shadow weights
    const float *bias;                 // forward only
    int M, NC, KC;                     // output columns, contraction length
    int act;                           // forward: activation; dgrad: activation whose derivative multiplies (uses `aux`)
    const bf16 *aux; int64_t ldaux;    // forward: residual;  dgrad: saved activation (post-activation value)
    const bf16 *aux2; int64_t ldaux2;  // dgrad: gradient arriving through the residual branch
    bf16 *C; int64_t ldc;
    bf16 *C2;                          // forward: optional pre-activation copy (same ld as C)
    int xcd_remap;
};

template <int BMW, int BK, int NS, int NW_ = 4>
struct Cfg {
    static constexpr int NW = NW_, BM = NW * BMW, BN = 128;
    static constexpr int CH = BK / 8;                        // 16-byte chunks per k-contiguous row
    static constexpr int RPI = 1024 / (BK * 2);              // rows per 1 KiB wave-instruction
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int A_DMA = A_BYTES / 1024 / NW, B_DMA = B_BYTES / 1024 / NW;
    static constexpr int PER_STAGE = A_DMA + B_DMA;
    static_assert(BK == 32 || BK == 64, "stage depth");
    static_assert(A_DMA * NW * 1024 == A_BYTES && B_DMA * NW * 1024 == B_BYTES, "stage must split evenly over the waves");
    static_assert(PER_STAGE * (NS - 1) < 64, "vmcnt is a 6-bit counter");
};

// k-contiguous image [rows][BK]: chunk c of row r lands in slot c ^ swz(r).
// BK = 32 (64-byte rows, four rows per 256-byte bank row): a ds_read_b128 is served in four 16-lane groups that are NOT
// contiguous -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- i.e. a group holds
// fragment rows 0-3 and 12-15 of k chunk g and rows 4-11 of k chunk g ^ 1.  Rows r, r+4, r+8, r+12 share their four
// slots, so their chunks must differ: with the row-quad's bit 1 as the XOR, (g, g^1, g^1, g) becomes (g, g^1, g^1^2, g^2).
// (The first version used (r >> 2) & 3, derived for contiguous 16-lane groups: 2-way on every fragment read,
// SQ_LDS_BANK_CONFLICT 0.38 - 0.45 of the LDS-active cycles.)
template <int BK>
__device__ __forceinline__ int swz(int r) { return BK == 32 ? ((r >> 2) & 2) : ((r >> 1) & 7); }
// natural [n rows][256-byte rows of k_in] image read by ds_read_b64_tr_b16 (two 32-lane groups): lanes l and l + 16 read
// rows 8 apart at the same columns -> bit 3 of the row goes into the chunk XOR as well (opposite 128-byte halves)
__device__ __forceinline__ int swz_nat(int r) { return (2 * (r & 7)) ^ (r & 8); }

template <int BMW, int BK, int NS, bool DGRAD, int NW>
__device__ __forceinline__ void issue_stage(const GdArgs &a, unsigned char *slot, int m0, int n0, int k0, int wave, int lane) {
    using C = Cfg<BMW, BK, NS, NW>;
#pragma unroll
    for (int i = 0; i < C::A_DMA; ++i) {
        const int inst = wave * C::A_DMA + i;
        const int r = inst * C::RPI + lane / C::CH, s = lane % C::CH;
        const int c = s ^ swz<BK>(r);
        dma16(a.A + (int64_t)min(m0 + r, a.M - 1) * a.lda + k0 + c * 8, slot + inst * 1024);
    }
    unsigned char *bs = slot + C::A_BYTES;
    if (!DGRAD) {       // W rows n0.. (output columns), k-contiguous: same image shape as A
#pragma unroll
        for (int i = 0; i < C::B_DMA; ++i) {
            const int inst = wave * C::B_DMA + i;
            const int n = inst * C::RPI + lane / C::CH, s = lane % C::CH;
            const int c = s ^ swz<BK>(n);
            dma16(a.W + (int64_t)(n0 + n) * a.ldw + k0 + c * 8, bs + inst * 1024);
        }
    } else {            // natural [BK rows n][128 cols k_in] image, 256-byte rows; chunk c of row r lands in slot c ^ swz_nat(r)
#pragma unroll
        for (int i = 0; i < C::B_DMA; ++i) {
            const int inst = wave * C::B_DMA + i;
            const int r = inst * 4 + (lane >> 4), s = lane & 15;
            const int c = s ^ swz_nat(r);
            dma16(a.W + (int64_t)(k0 + r) * a.ldw + n0 + c * 8, bs + inst * 1024);
        }
    }
}

// B fragment of the dgrad: k = rows (mr + q | mr + 4 + q supplied per lane), lane index = column colbase + (lane & 15)
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char *img, int mr, int colbase, int q, int p) {
    const int col = colbase + 4 * p;
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const int r0 = mr + q, r1 = mr + 4 + q;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(img + r0 * 256 + (((col >> 3) ^ swz_nat(r0)) * 16) + (col & 7) * 2));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(img + r1 * 256 + (((col >> 3) ^ swz_nat(r1)) * 16) + (col & 7) * 2));
    struct { s16x4 a, b; } pr = {lo, hi};
    return __builtin_bit_cast(bf16x8, pr);
}

template <int BMW, int BK, int NS, bool DGRAD, int NW = 4>
__global__ __launch_bounds__(NW * 64, (BMW == 64 && NW == 4 ? 2 : 1)) void gemm_dma_kernel(const GdArgs a) {
    using C = Cfg<BMW, BK, NS, NW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TM = BMW / 16, TN = 8, D = NS - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cq = lane & 15, q = cq >> 2, p = cq & 3;
    // XCD-aware placement: all column blocks of one row tile run on the SAME XCD, back to back (the A tile is fetched into
    // that L2 once instead of once per column block; the W column blocks, 128 KB each at K = 512, stay L2-resident).
    // Dispatch order is blockIdx.x fastest, dealt round-robin over the 8 XCDs.
    int bx = blockIdx.x, by = blockIdx.y;
    if (a.xcd_remap && gridDim.y > 1) {
        const int T = gridDim.y, gx = gridDim.x, lin = blockIdx.y * gx + blockIdx.x, G = 8 * T, ngrp = gx / 8;
        if (lin < ngrp * G) { by = (lin % G) / 8; bx = (lin / G) * 8 + lin % 8; }
        else { const int idx = lin - ngrp * G, rem = gx - 8 * ngrp; bx = 8 * ngrp + idx % rem; by = idx / rem; }
    }
    const int m0 = bx * C::BM, n0 = by * C::BN;
    const int nst = a.KC / BK;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < nst) issue_stage<BMW, BK, NS, DGRAD, NW>(a, smem + s * C::STAGE, m0, n0, s * BK, wave, lane);
    for (int s = 0; s < nst; ++s) {
        if (s + D - 1 < nst) wait_vm<C::PER_STAGE * (D - 1)>();
        else wait_vm<0>();
        barrier();
        if (s + D < nst) issue_stage<BMW, BK, NS, DGRAD, NW>(a, smem + ((s + D) % NS) * C::STAGE, m0, n0, (s + D) * BK, wave, lane);
        const unsigned char *As = smem + (s % NS) * C::STAGE, *Bs = As + C::A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 af[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wave * BMW + i * 16 + cq;
                af[i] = *reinterpret_cast<const bf16x8 *>(As + r * (BK * 2) + (((kk * 4 + g) ^ swz<BK>(r)) * 16));
            }
            if (BMW == 64) {
                // 64-row wave tile (two waves per SIMD): request the whole step's B fragments before the first MFMA --
                // left alone the scheduler fetches them two at a time and waits for each pair with the matrix pipe idle
                bf16x8 bf[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!DGRAD) {
                        const int n = j * 16 + cq;
                        bf[j] = *reinterpret_cast<const bf16x8 *>(Bs + n * (BK * 2) + (((kk * 4 + g) ^ swz<BK>(n)) * 16));
                    } else {
                        bf[j] = tr_frag(Bs, kk * 32 + 8 * g, j * 16, q, p);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] = mma(af[i], bf[j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bf16x8 bf;
                if (!DGRAD) {
                    const int n = j * 16 + cq;
                    bf = *reinterpret_cast<const bf16x8 *>(Bs + n * (BK * 2) + (((kk * 4 + g) ^ swz<BK>(n)) * 16));
                } else {
                    bf = tr_frag(Bs, kk * 32 + 8 * g, j * 16, q, p);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = mma(af[i], bf, acc[i][j]);
            }
        }
    }
    // ---- epilogue: 16 rows x 128 columns per pass through this wave's fp32 scratch ----
    barrier();                                     // every wave is done with the operand ring
    constexpr int LDC = C::BN + 4;
    float *Cs = reinterpret_cast<float *>(smem) + wave * 16 * LDC;
    const bool has_bias = !DGRAD && a.bias != nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int lc = j * 16 + cq;
            const float bv = has_bias ? a.bias[n0 + lc] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(4 * g + r) * LDC + lc] = acc[i][j][r] + bv;
        }
        wave_lds_fence();
        for (int c = lane; c < 16 * (C::BN / 8); c += 64) {
            const int lr = c / (C::BN / 8), lc = (c % (C::BN / 8)) * 8;
            const int row = m0 + wave * BMW + i * 16 + lr;
            if (row >= a.M) continue;
            float v[8];
            *reinterpret_cast<float4 *>(v) = *reinterpret_cast<const float4 *>(Cs + lr * LDC + lc);
            *reinterpret_cast<float4 *>(v + 4) = *reinterpret_cast<const float4 *>(Cs + lr * LDC + lc + 4);
            const int64_t col = n0 + lc;
            if (!DGRAD) {
                if (a.C2) store16(a.C2 + (int64_t)row * a.ldc + col, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = act_fwd(a.act, v[e]);
                if (a.aux) {
                    float d[8];
                    load16(a.aux + (int64_t)row * a.ldaux + col, d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += d[e];
                }
            } else {
                if (a.aux) {
                    float d[8];
                    load16(a.aux + (int64_t)row * a.ldaux + col, d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= act_bwd(a.act, d[e]);
                }
                if (a.aux2) {
                    float d[8];
                    load16(a.aux2 + (int64_t)row * a.ldaux2 + col, d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += d[e];
                }
            }
            store16(a.C + (int64_t)row * a.ldc + col, v);
        }
    }
}

template <int BMW, int BK, int NS, bool DGRAD, int NW = 4>
int launch_t(const GdArgs &a, hipStream_t s) {
    using C = Cfg<BMW, BK, NS, NW>;
    const size_t ring = (size_t)NS * C::STAGE, scratch = (size_t)NW * 16 * (C::BN + 4) * 4;
    const size_t bytes = ring > scratch ? ring : scratch;
    auto kern = gemm_dma_kernel<BMW, BK, NS, DGRAD, NW>;
    MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    ProfScope prof(s);
    hipLaunchKernelGGL(kern, dim3(ceil_div(a.M, C::BM), a.NC / C::BN), dim3(NW * 64), bytes, s, a);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================================
// 256 x 256 workgroup tile, 2 x 2 waves of 128 x 128, TRANSPOSED product with a direct epilogue.
//   * per 32-deep k step a wave reads 8 + 8 fragments (16 KB) for 64 MFMAs = 16 B/clk at full MFMA rate, 64 B/clk for the
//     four SIMDs, half the LDS port (160 B/clk, over the port, for the 32 x 128 wave tile above); 256 accumulator
//     registers (AGPRs) = one wave per SIMD, all 16 fragments of a step requested before its first MFMA;
//   * measured on the first version of this kernel (config-4 QKV layer, 133120 x 1536 x 512): main loop alone 0.113 ms
//     (1.85 PFLOP/s), with the fp32-through-LDS epilogue of the kernel above 0.56 ms -- at K = 512 the epilogue, not the
//     k loop, was the cost.  So the MFMA computes C^T = W X^T: the accumulator tile then has lane = output ROW and
//     registers = 4 output columns, and with the W rows dealt to the MFMA row slots as n = 32 * (slot >> 2) + 4 * j +
//     (slot & 3) (tile j, slot 0..15) lane group g ends up holding the 32 CONSECUTIVE columns 32 g .. 32 g + 31 of its row:
//     bias / activation / residual / store run straight from the accumulators, four 16-byte stores per lane per row
//     tile, 256 contiguous bytes per row -- no LDS, no barrier.
//   * LDS images: X [256][32] as above; W rows carry the chunk swizzle (n >> 5) & 2 so that the permuted row set of one
//     fragment read (4 runs of 4 rows, 32 rows apart) spreads over all banks (same lane-group argument as swz<32>); dgrad: natural [32 n][256 k_in] image,
//     chunk swizzle (n & 3), read by ds_read_b64_tr_b16 with the four 4-column chunks of a read placed 32 columns apart.
// =================================================================================================================
template <int WM, int WN, int TM, int NS>
struct BigCfg {
    static constexpr int NW = WM * WN, BM = WM * TM * 16, BN = WN * 128, BK = 32;
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    static constexpr int A_DMA = A_BYTES / 1024 / NW, B_DMA = B_BYTES / 1024 / NW, PER_STAGE = A_DMA + B_DMA;
    static_assert(A_DMA * NW * 1024 == A_BYTES && B_DMA * NW * 1024 == B_BYTES, "stage must split evenly over the waves");
    static_assert(PER_STAGE * (NS - 1) < 64, "vmcnt is a 6-bit counter");
    static_assert(NS * STAGE <= 160 * 1024, "LDS");
};

template <class C, bool DGRAD>
__device__ __forceinline__ void big_issue(const GdArgs &a, unsigned char *slot, int m0, int n0, int k0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < C::A_DMA; ++i) {              // 16 rows of 64 B per wave-instruction
        const int inst = wave * C::A_DMA + i;
        const int r = inst * 16 + (lane >> 2), s = lane & 3;
        dma16(a.A + (int64_t)min(m0 + r, a.M - 1) * a.lda + k0 + (s ^ swz<32>(r)) * 8, slot + inst * 1024);
    }
    unsigned char *bs = slot + C::A_BYTES;
    if (!DGRAD) {
#pragma unroll
        for (int i = 0; i < C::B_DMA; ++i) {
            const int inst = wave * C::B_DMA + i;
            const int n = inst * 16 + (lane >> 2), s = lane & 3;
            dma16(a.W + (int64_t)(n0 + n) * a.ldw + k0 + (s ^ ((n >> 5) & 2)) * 8, bs + inst * 1024);
        }
    } else {            // natural [32 rows n][BN cols k_in] image; chunk c of row r lands in slot c ^ (r & 3)
        constexpr int CPR = C::BN / 8, RPI = 64 / CPR;       // 16-byte chunks per row, rows per wave-instruction
#pragma unroll
        for (int i = 0; i < C::B_DMA; ++i) {
            const int inst = wave * C::B_DMA + i;
            const int r = inst * RPI + lane / CPR, s = lane % CPR;
            dma16(a.W + (int64_t)(k0 + r) * a.ldw + n0 + (s ^ (r & 3)) * 8, bs + inst * 1024);
        }
    }
}

// one 32-deep k step of a wave: all 16 fragments requested before the first MFMA, then TM x 8 MFMAs
template <class C, int TM, bool DGRAD>
__device__ __forceinline__ void big_step(const unsigned char *As, const unsigned char *Bs, f32x4 (&acc)[TM][8], int wm, int wn, int lane) {
    constexpr int TN = 8, ROWB = C::BN * 2;
    const int g = lane >> 4, cq = lane & 15, q = cq >> 2, p = cq & 3;
    bf16x8 xf[TM], wf[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = (wm * TM + i) * 16 + cq;
        xf[i] = *reinterpret_cast<const bf16x8 *>(As + r * 64 + ((g ^ swz<32>(r)) * 16));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        if (!DGRAD) {
            const int n = wn * 128 + 32 * q + 4 * j + p;                   // (n >> 5) & 3 == q
            wf[j] = *reinterpret_cast<const bf16x8 *>(Bs + n * 64 + ((g ^ (q & 2)) * 16));
        } else {
            // lane (q, p) of a transposing read supplies row q of the k quad, 4 columns of chunk p; chunk p sits at
            // columns 32 p + 4 j: after the transpose lane cq owns column 32 (cq >> 2) + 4 j + (cq & 3)
            typedef __attribute__((address_space(3))) s16x4 lds_v4;
            const int col = wn * 128 + 32 * p + 4 * j;
            const int r0 = 8 * g + q, r1 = r0 + 4;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(Bs + r0 * ROWB + (((col >> 3) ^ (r0 & 3)) * 16) + (col & 7) * 2));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(Bs + r1 * ROWB + (((col >> 3) ^ (r1 & 3)) * 16) + (col & 7) * 2));
            struct { s16x4 a, b; } pr = {lo, hi};
            wf[j] = __builtin_bit_cast(bf16x8, pr);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mma(wf[j], xf[i], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
}

// epilogue straight from the accumulators: lane = row, 32 consecutive columns.  The activation is a compile-time constant
// of each copy: with the runtime switch inlined per value the unrolled epilogue was 27 k instructions of compare-and-branch.
template <int TM, bool DGRAD>
__device__ __forceinline__ void big_epilogue(const GdArgs &a, f32x4 (&acc)[TM][8], int row0, int64_t col0, int cq) {
    float bv[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) bv[t] = (!DGRAD && a.bias) ? a.bias[col0 + t] : 0.f;
    auto epilogue = [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = row0 + i * 16 + cq;
            if (row >= a.M) continue;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[i][2 * c + (e >> 2)][e & 3] + bv[8 * c + e];
                const int64_t col = col0 + 8 * c;
                if (!DGRAD) {
                    if (a.C2) store16(a.C2 + (int64_t)row * a.ldc + col, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = act_fwd(ACT, v[e]);
                    if (a.aux) {
                        float d[8];
                        load16(a.aux + (int64_t)row * a.ldaux + col, d);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += d[e];
                    }
                } else {
                    if (ACT != MIVIT_ACT_NONE) {
                        float d[8];
                        load16(a.aux + (int64_t)row * a.ldaux + col, d);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= act_bwd(ACT, d[e]);
                    }
                    if (a.aux2) {
                        float d[8];
                        load16(a.aux2 + (int64_t)row * a.ldaux2 + col, d);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += d[e];
                    }
                }
                store16(a.C + (int64_t)row * a.ldc + col, v);
            }
        }
    };
    const int act = (DGRAD && !a.aux) ? (int)MIVIT_ACT_NONE : a.act;
    switch (act) {
        case MIVIT_ACT_RELU: epilogue(std::integral_constant<int, MIVIT_ACT_RELU>{}); break;
        case MIVIT_ACT_LEAKY_RELU: epilogue(std::integral_constant<int, MIVIT_ACT_LEAKY_RELU>{}); break;
        case MIVIT_ACT_GELU: epilogue(std::integral_constant<int, MIVIT_ACT_GELU>{}); break;
        default: epilogue(std::integral_constant<int, MIVIT_ACT_NONE>{}); break;
    }
}

template <int WM, int WN, int TM>
constexpr int big_min_blocks() { return (TM <= 4 ? 8 : 4) / (WM * WN) > 0 ? (TM <= 4 ? 8 : 4) / (WM * WN) : 1; }

// one workgroup per output tile
template <int WM, int WN, int TM, int NS, bool DGRAD>
__global__ __launch_bounds__(WM * WN * 64, (big_min_blocks<WM, WN, TM>())) void gemm_big_kernel(const GdArgs a) {
    using C = BigCfg<WM, WN, TM, NS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int D = NS - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cq = lane & 15;
    const int wm = wave / WN, wn = wave % WN;
    int bx = blockIdx.x, by = blockIdx.y;
    if (a.xcd_remap && gridDim.y > 1) {        // as above: the column blocks of one row tile on one XCD
        const int T = gridDim.y, gx = gridDim.x, lin = blockIdx.y * gx + blockIdx.x, G = 8 * T, ngrp = gx / 8;
        if (lin < ngrp * G) { by = (lin % G) / 8; bx = (lin / G) * 8 + lin % 8; }
        else { const int idx = lin - ngrp * G, rem = gx - 8 * ngrp; bx = 8 * ngrp + idx % rem; by = idx / rem; }
    }
    const int m0 = bx * C::BM, n0 = by * C::BN;
    const int nst = a.KC / 32;

    f32x4 acc[TM][8];                        // acc[i][j][r]: row m0 + (wm*TM + i)*16 + cq, column n0 + wn*128 + 32*g + 4*j + r
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < nst) big_issue<C, DGRAD>(a, smem + s * C::STAGE, m0, n0, s * 32, wave, lane);
    for (int s = 0; s < nst; ++s) {
        if (s + D - 1 < nst) wait_vm<C::PER_STAGE * (D - 1)>();
        else wait_vm<0>();
        barrier();
        if (s + D < nst) big_issue<C, DGRAD>(a, smem + ((s + D) % NS) * C::STAGE, m0, n0, (s + D) * 32, wave, lane);
        const unsigned char *As = smem + (s % NS) * C::STAGE;
        big_step<C, TM, DGRAD>(As, As + C::A_BYTES, acc, wm, wn, lane);
    }
    big_epilogue<TM, DGRAD>(a, acc, m0 + wm * TM * 16, (int64_t)n0 + wn * 128 + 32 * g, cq);
}

// =================================================================================================================
// Persistent form of the same kernel.  Measured on the one-tile-per-workgroup version (QKV layer, 0.083 ms at MFMA peak):
// MFMAs + barriers alone 0.13 - 0.17 ms, + LDS-DMA 0.21 - 0.26, + epilogue 0.32 - 0.41: at K = 512 a tile is only 16
// stages (3.4 - 6.8 us of MFMA work), so what a workgroup pays per tile -- launch, filling the ring from cold, draining
// it, the epilogue -- is as large as the k loop itself.  Here a workgroup stays resident and walks a list of tiles with
// ONE stage ring running across tile boundaries: the first D stages of the next tile are requested during the last D
// stages of the current one, and they land while the epilogue of the current tile runs (two workgroups per CU at TM = 4:
// the other one's MFMAs fill the epilogue).  Tile order: tile t = x + 8 u belongs to XCD x (workgroup index mod 8), and u
// walks the column blocks of one row tile first, so the CUs of an XCD work on the column blocks of the same few row tiles
// at the same time (the A tile enters that L2 once).
// =================================================================================================================
template <int WM, int WN, int TM, int NS, bool DGRAD>
__global__ __launch_bounds__(WM * WN * 64, (big_min_blocks<WM, WN, TM>())) void gemm_pers_kernel(const GdArgs a) {
    using C = BigCfg<WM, WN, TM, NS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int D = NS - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cq = lane & 15;
    const int wm = wave / WN, wn = wave % WN;
    const int nst = a.KC / 32;
    const int nrt = (a.M + C::BM - 1) / C::BM, ncb = a.NC / C::BN, ntile = ((nrt + 7) / 8) * 8 * ncb;
    auto coords = [&](int t, int &m0, int &n0) { const int x = t & 7, u = t >> 3; m0 = (x + 8 * (u / ncb)) * C::BM; n0 = (u % ncb) * C::BN; };
    auto next_tile = [&](int t) {            // next tile of this workgroup whose row tile exists (padding tiles: last group of 8)
        for (t += gridDim.x; t < ntile; t += gridDim.x) { int m0, n0; coords(t, m0, n0); if (m0 < a.M) return t; }
        return -1;
    };
    int t = next_tile((int)blockIdx.x - (int)gridDim.x);
    if (t < 0) return;
    int tn = next_tile(t);
    int m0, n0, m0n = 0, n0n = 0;
    coords(t, m0, n0);
    if (tn >= 0) coords(tn, m0n, n0n);

    // stage G of the workgroup's whole run lives in ring slot G % NS; `slot` tracks it without a division
    int slot_c = 0, slot_i = 0;              // slot of the stage consumed next / issued next
    auto issue = [&](int mm, int nn, int k0) {
        big_issue<C, DGRAD>(a, smem + slot_i * C::STAGE, mm, nn, k0, wave, lane);
        slot_i = slot_i + 1 == NS ? 0 : slot_i + 1;
    };
#pragma unroll
    for (int s = 0; s < D; ++s) {            // nst >= 4 > D (gemm_dma_supported: K >= 128)
        if (s < nst) issue(m0, n0, s * 32);
    }
    while (true) {
        f32x4 acc[TM][8];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < nst; ++s) {
            const bool more = s + D < nst || tn >= 0;            // a stage is issued in this iteration
            // s == 0 of a later tile: the queue also holds the previous epilogue's loads / stores -> drain it (the ring
            // stages ahead were requested D stages ago and have landed); last stages of the run: nothing younger follows
            if (s == 0 || !(s + D - 1 < nst || tn >= 0)) wait_vm<0>();
            else wait_vm<C::PER_STAGE * (D - 1)>();
            barrier();
            if (s + D < nst) issue(m0, n0, (s + D) * 32);
            else if (tn >= 0) issue(m0n, n0n, (s + D - nst) * 32);
            (void)more;
            const unsigned char *As = smem + slot_c * C::STAGE;
            slot_c = slot_c + 1 == NS ? 0 : slot_c + 1;
            big_step<C, TM, DGRAD>(As, As + C::A_BYTES, acc, wm, wn, lane);
        }
        big_epilogue<TM, DGRAD>(a, acc, m0 + wm * TM * 16, (int64_t)n0 + wn * 128 + 32 * g, cq);
        if (tn < 0) break;
        t = tn; m0 = m0n; n0 = n0n;
        tn = next_tile(t);
        if (tn >= 0) coords(tn, m0n, n0n);
    }
}

template <int WM, int WN, int TM, int NS, bool DGRAD>
int launch_big(const GdArgs &a, hipStream_t s) {
    using C = BigCfg<WM, WN, TM, NS>;
    const size_t bytes = (size_t)NS * C::STAGE;
    auto kern = gemm_big_kernel<WM, WN, TM, NS, DGRAD>;
    MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    ProfScope prof(s);
    hipLaunchKernelGGL(kern, dim3(ceil_div(a.M, C::BM), a.NC / C::BN), dim3(C::NW * 64), bytes, s, a);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

template <int WM, int WN, int TM, int NS, bool DGRAD>
int launch_pers(const GdArgs &a, hipStream_t s) {
    using C = BigCfg<WM, WN, TM, NS>;
    const size_t bytes = (size_t)NS * C::STAGE;
    auto kern = gemm_pers_kernel<WM, WN, TM, NS, DGRAD>;
    MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    const int per_cu = (int)((size_t)160 * 1024 / bytes) < big_min_blocks<WM, WN, TM>() ? (int)((size_t)160 * 1024 / bytes) : big_min_blocks<WM, WN, TM>();
    const int ntile = ceil_div(ceil_div(a.M, C::BM), 8) * 8 * (a.NC / C::BN);
    const int resident = 256 * (per_cu < 1 ? 1 : per_cu);
    ProfScope prof(s);
    hipLaunchKernelGGL(kern, dim3(ntile < resident ? ntile : resident), dim3(C::NW * 64), bytes, s, a);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

// tile variants, selectable for the sizing sweep: MIVIT_GEMM_DMA_VARIANT / mivit_gemm_dma_set_variant (0 = default)
int g_variant = getenv("MIVIT_GEMM_DMA_VARIANT") ? atoi(getenv("MIVIT_GEMM_DMA_VARIANT")) : 0;

template <bool DGRAD>
int launch_variant(GdArgs a, hipStream_t s) {
    static const int remap = getenv("MIVIT_XCD_REMAP") ? atoi(getenv("MIVIT_XCD_REMAP")) : 1;
    a.xcd_remap = remap;
    const int variant = g_variant;
    // transposed-product kernels (direct epilogue): 20..27 one workgroup per tile, 30..37 persistent; 19 = first generation
    if (variant >= 20 && a.M >= 256) {
        switch (variant) {
            case 20: if (a.NC % 256 == 0) return launch_big<2, 2, 8, 4, DGRAD>(a, s); break;     // 256 x 256, 1 wave / SIMD
            case 21: return launch_big<4, 1, 4, 3, DGRAD>(a, s);                                 // 256 x 128, 4 waves of 64 x 128, 2 workgroups / CU
            case 22: if (a.NC % 256 == 0) return launch_big<2, 2, 4, 3, DGRAD>(a, s); break;     // 128 x 256, 4 waves of 64 x 128, 2 workgroups / CU
            case 23: return launch_big<8, 1, 2, 3, DGRAD>(a, s);                                 // 256 x 128, 8 waves of 32 x 128
            case 27: if (a.NC % 256 == 0) return launch_big<4, 2, 4, 3, DGRAD>(a, s); break;     // 256 x 256, 8 waves of 64 x 128
            case 30: if (a.NC % 256 == 0) return launch_pers<2, 2, 8, 4, DGRAD>(a, s); break;
            case 31: return launch_pers<4, 1, 4, 3, DGRAD>(a, s);
            case 32: if (a.NC % 256 == 0) return launch_pers<2, 2, 4, 3, DGRAD>(a, s); break;
            case 33: return launch_pers<8, 1, 2, 3, DGRAD>(a, s);
            case 37: if (a.NC % 256 == 0) return launch_pers<4, 2, 4, 3, DGRAD>(a, s); break;
            default: break;
        }
    }
    switch (variant) {
        case 1: return launch_t<64, 32, 3, DGRAD>(a, s);     // 256 x 128 tile, 72 KB ring: 1 wave / SIMD (268 registers)
        case 2: return launch_t<64, 32, 4, DGRAD>(a, s);
        case 3: return launch_t<64, 64, 2, DGRAD>(a, s);
        case 4: return launch_t<32, 64, 3, DGRAD>(a, s);     // 128 x 128 tile, 96 KB ring: 1 workgroup / CU
        case 5: return launch_t<32, 64, 2, DGRAD>(a, s);     // 64 KB ring: 2 workgroups / CU, 32 MFMAs per barrier
        case 6: return launch_t<32, 32, 3, DGRAD, 8>(a, s);  // 256 x 128 tile, 8 waves, 72 KB ring
        case 7: return launch_t<32, 32, 2, DGRAD, 8>(a, s);  // 256 x 128 tile, 8 waves, 48 KB ring
        case 8: return launch_t<32, 64, 2, DGRAD, 8>(a, s);  // 256 x 128 tile, 8 waves, 96 KB ring
        case 9: return launch_t<32, 32, 3, DGRAD>(a, s);
        case 10: return launch_t<64, 32, 2, DGRAD>(a, s);    // 256 x 128 tile, 64-row waves, 48 KB ring: 2 workgroups / CU by registers
        case 11: return launch_t<64, 32, 2, DGRAD, 8>(a, s); // 512 x 128 tile, 8 waves of 64 x 128, 80 KB ring     // 128 x 128 tile, 4 waves, 48 KB ring: 3 workgroups / CU
        default:                                             // 256 x 128 tile, 8 waves of 32 x 128 (measured best: c4 forward
            return DGRAD ? launch_t<32, 32, 2, true, 8>(a, s)    //   4.17 -> 3.75 ms with 3 slots, dgrad 4.05 -> 3.69 ms with 2)
                         : launch_t<32, 32, 3, false, 8>(a, s);
    }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// wide-layer shapes only: the row-stream kernels own K in {128, 256, 384}
bool gemm_dma_supported(int M, int NC, int KC, bool dgrad) {
    static const bool off = getenv("MIVIT_NO_GEMM_DMA") != nullptr;
    if (off || M < 256 || NC % 128 != 0 || KC % 64 != 0 || KC < 128) return false;
    return true;
}

int launch_gemm_dma_fwd(const void *x, int64_t ldx, const void *W_bf16, const float *bias, int M, int N, int K, int act,
                        const void *resid, int64_t ldr, void *y, int64_t ldy, void *y_preact, hipStream_t s) {
    MIVIT_CHECK(aligned16(x) && aligned16(W_bf16) && aligned16(y) && aligned16(resid) && aligned16(y_preact) && ldx % 8 == 0 &&
                ldy % 8 == 0 && (!resid || ldr % 8 == 0), "gemm_dma_fwd: operands must be 16-byte aligned");
    GdArgs a = {};
    a.A = static_cast<const bf16 *>(x); a.lda = ldx; a.W = static_cast<const bf16 *>(W_bf16); a.ldw = K; a.bias = bias;
    a.M = M; a.NC = N; a.KC = K; a.act = act; a.aux = static_cast<const bf16 *>(resid); a.ldaux = ldr;
    a.C = static_cast<bf16 *>(y); a.ldc = ldy; a.C2 = static_cast<bf16 *>(y_preact);
    return launch_variant<false>(a, s);
}

// dx [M,K] = (dy [M,N] . W [N,K]) * act'(saved) + dres
int launch_gemm_dma_dgrad(const void *dy, int64_t lddy, const void *W_bf16, int M, int N, int K, int act, const void *saved,
                          int64_t lds, const void *dres, int64_t lddr, void *dx, int64_t lddx, hipStream_t s) {
    MIVIT_CHECK(aligned16(dy) && aligned16(W_bf16) && aligned16(dx) && aligned16(saved) && aligned16(dres) && lddy % 8 == 0 &&
                lddx % 8 == 0 && (!saved || lds % 8 == 0) && (!dres || lddr % 8 == 0), "gemm_dma_dgrad: operands must be 16-byte aligned");
    GdArgs a = {};
    a.A = static_cast<const bf16 *>(dy); a.lda = lddy; a.W = static_cast<const bf16 *>(W_bf16); a.ldw = K;
    a.M = M; a.NC = K; a.KC = N; a.act = act; a.aux = act != MIVIT_ACT_NONE ? static_cast<const bf16 *>(saved) : nullptr; a.ldaux = lds;
    a.aux2 = static_cast<const bf16 *>(dres); a.ldaux2 = lddr;
    a.C = static_cast<bf16 *>(dx); a.ldc = lddx;
    return launch_variant<true>(a, s);
}

extern "C" int mivit_gemm_dma_set_variant(int v) { const int old = g_variant; g_variant = v; return old; }
extern "C" int mivit_gemm_dma_supported(int M, int N, int K, int dgrad) {
    return dgrad ? gemm_dma_supported(M, K, N, true) : gemm_dma_supported(M, N, K, false);
}
extern "C" int mivit_gemm_dma_fwd(const void *x, int64_t ldx, const void *W_bf16, const float *bias, int M, int N, int K, int act,
                                  const void *resid, int64_t ldr, void *y, int64_t ldy, void *y_preact, void *stream) {
    MIVIT_CHECK(x && W_bf16 && y, "gemm_dma_fwd: null pointer");
    if (!gemm_dma_supported(M, N, K, false)) { mivit_set_error("gemm_dma_fwd: unsupported shape M=%d N=%d K=%d", M, N, K); return 3; }
    prof_set_tag(MIVIT_PROF_OP);
    return launch_gemm_dma_fwd(x, ldx, W_bf16, bias, M, N, K, act, resid, ldr, y, ldy, y_preact, static_cast<hipStream_t>(stream));
}
extern "C" int mivit_gemm_dma_dgrad(const void *dy, int64_t lddy, const void *W_bf16, int M, int N, int K, int act,
                                    const void *saved, int64_t lds, const void *dres, int64_t lddr, void *dx, int64_t lddx,
                                    void *stream) {
    MIVIT_CHECK(dy && W_bf16 && dx, "gemm_dma_dgrad: null pointer");
    if (!gemm_dma_supported(M, K, N, true)) { mivit_set_error("gemm_dma_dgrad: unsupported shape M=%d N=%d K=%d", M, N, K); return 3; }
    prof_set_tag(MIVIT_PROF_OP);
    return launch_gemm_dma_dgrad(dy, lddy, W_bf16, M, N, K, act, saved, lds, dres, lddr, dx, lddx, static_cast<hipStream_t>(stream));
}
