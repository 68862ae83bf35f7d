// Weight gradients of the encoder-layer projections in bf16 mode:  dW[N, K] = dY[M, N]^T . X[M, K]  (M = all tokens).
// Same streaming structure as the embedding weight gradient (embed.hip): both operands are natural [m][.] bf16 tiles
// brought in by LDS-DMA into a ring, the contraction over rows m uses ds_read_b64_tr_b16 for BOTH fragments, the
// reduction is split over row ranges into fp32 slabs that slab_reduce sums in a fixed order (deterministic).
#include "common.h"
#include "stream_prims.h"
#include <stdlib.h>
#include "elem.h"          // element type of this translation unit (bf16, or f16 under -DMIVIT_ELEM_F16): after every other include

namespace {



struct WgArgs {
    const bf16 *dY; int64_t lddy; const bf16 *X; int64_t ldx; float *slabs;
    int M, N, K, rows_per_split;
    float *bias_part;      // optional [splits][N]: column sums of dY (bias gradient), taken from the LDS image
    int xcd_remap;
};

constexpr int TILE = 128;        // output tile; BMR reduction rows per stage, NS ring slots are template parameters
constexpr int BMR_MAX = 64;

// natural [BMR rows m][128 cols] bf16 image, 256-byte rows; chunk c of row r lands in slot c ^ (2 * (r & 7))
template <int BMR>
__device__ __forceinline__ void issue_img(const bf16 *src, int64_t ld, int col0, int mrow, int mend, unsigned char *img,
                                          int wave, int lane) {
    constexpr int DMA_PER_WAVE = BMR * TILE * 2 / 1024 / 4;
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int inst = wave * DMA_PER_WAVE + i;
        const int r = inst * 4 + (lane >> 4), s = lane & 15;
        const int c = s ^ (2 * (r & 7));
        const int gm = min(mrow + r, mend - 1);
        dma16(src + (int64_t)gm * ld + col0 + c * 8, img + inst * 1024);
    }
}

// fragment with k = rows (mr + 0..3 | mr + 4 + 0..3) and lane index = column col0 + (lane & 15)
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char *img, int mr, int colbase, int q, int p, int valid) {
    const int col = colbase + 4 * p;
    s16x4 lo, hi;
    {
        const int r = mr + q;
        const int c = (col >> 3) ^ (2 * (r & 7));
        lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4 *)(img + r * 256 + c * 16 + (col & 7) * 2));
    }
    {
        const int r = mr + 4 + q;
        const int c = (col >> 3) ^ (2 * (r & 7));
        hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4 *)(img + r * 256 + c * 16 + (col & 7) * 2));
    }
    // rows past the end of the split repeat the last row (clamped DMA): zero those k slots (the lane RECEIVES rows
    // mr + 0..3 in lo[0..3] and mr + 4..7 in hi[0..3], whatever address it supplied)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (mr + e >= valid) lo[e] = 0;
        if (mr + 4 + e >= valid) hi[e] = 0;
    }
    struct { s16x4 a, b; } pr = {lo, hi};
    return __builtin_bit_cast(bf16x8, pr);
}

template <int NS, int BMR>
__global__ __launch_bounds__(256) void wgrad_dma_kernel(const WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int IMG = BMR * TILE * 2, STAGE = 2 * IMG, DMA_PER_WAVE = IMG / 1024 / 4;
    constexpr int TM = 4, TN = 4, D = NS - 1, PER_STAGE = 2 * DMA_PER_WAVE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware placement (workgroups are dealt round-robin over the 8 XCDs in dispatch order, x fastest): the T output
    // tiles of one row split read the same dY / X rows, so within 8 * T consecutive workgroups XCD k hosts all tiles of
    // split 8 * group + k
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int T = gridDim.x * gridDim.y, lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, G = 8 * T;
        if (a.xcd_remap && T > 1 && lin < (int)(T * gridDim.z) / G * G) {
            const int t = (lin % G) / 8;
            bz = (lin / G) * 8 + lin % 8; bx = t % gridDim.x; by = t / gridDim.x;
        }
    }
    const int k0 = bx * TILE, n0 = by * TILE;
    const int mb = bz * a.rows_per_split, me = min(a.M, mb + a.rows_per_split);
    const int nst = (me - mb + BMR - 1) / BMR;
    const int g = lane >> 4, cq = lane & 15, q = cq >> 2, p = cq & 3;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = a.bias_part != nullptr && bx == 0;      // one k-tile column of blocks sums dY
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < nst) {
            issue_img<BMR>(a.dY, a.lddy, n0, mb + s * BMR, me, smem + s * STAGE, wave, lane);
            issue_img<BMR>(a.X, a.ldx, k0, mb + s * BMR, me, smem + s * STAGE + IMG, wave, lane);
        }
    for (int s = 0; s < nst; ++s) {
        if (s + D - 1 < nst) wait_vm<PER_STAGE *(D - 1)>();
        else wait_vm<0>();
        barrier();
        if (s + D < nst) {
            unsigned char *slot = smem + ((s + D) % NS) * STAGE;
            issue_img<BMR>(a.dY, a.lddy, n0, mb + (s + D) * BMR, me, slot, wave, lane);
            issue_img<BMR>(a.X, a.ldx, k0, mb + (s + D) * BMR, me, slot + IMG, wave, lane);
        }
        const unsigned char *As = smem + (s % NS) * STAGE, *Bs = As + IMG;
        const int valid = min(BMR, me - (mb + s * BMR));
        if (do_bias) {      // thread = (16-byte chunk tid & 15 = 8 columns, row lane tid >> 4): rows lane, lane + 16, ...
            const int ch = tid & 15;
#pragma unroll
            for (int i = 0; i < BMR / 16; ++i) {
                const int r = (tid >> 4) + 16 * i;
                float v[8];
                load16(reinterpret_cast<const bf16 *>(As + r * 256 + ((ch ^ (2 * (r & 7))) << 4)), v);
                if (r < valid)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[e] += v[e];
            }
        }
#pragma unroll
        for (int kk = 0; kk < BMR / 32; ++kk) {
            const int mr = kk * 32 + 8 * g;
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = tr_frag(As, mr, wm * 64 + i * 16, q, p, valid);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = tr_frag(Bs, mr, wn * 64 + j * 16, q, p, BMR);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mma(af[i], bf[j], acc[i][j]);
        }
    }
    if (do_bias) {
        barrier();
        float *red = reinterpret_cast<float *>(smem);           // [16 row lanes][128 columns]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 4) * 128 + (tid & 15) * 8 + e] = bsum[e];
        barrier();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int y = 0; y < 16; ++y) t += red[y * 128 + tid];
            a.bias_part[(int64_t)bz * a.N + n0 + tid] = t;
        }
    }
    float *out = a.slabs + (int64_t)bz * a.N * a.K;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wm * 64 + i * 16 + 4 * g + r;
                out[(int64_t)n * a.K + k0 + wn * 64 + j * 16 + cq] = acc[i][j][r];
            }
}

int dma_splits(int M, int N, int K) {
    const long tiles = (long)(N / TILE) * (K / TILE);
    static const int target = getenv("MIVIT_WGRAD_DMA_BLOCKS") ? atoi(getenv("MIVIT_WGRAD_DMA_BLOCKS")) : 512;
    long s = (target + tiles - 1) / tiles;       // two resident workgroups per CU, one round
    const long maxs = (M + 127) / 128;           // at least two 64-row stages per split
    if (s > maxs) s = maxs;
    return s < 1 ? 1 : (int)s;
}

}  // namespace

bool wgrad_dma_supported(int M, int N, int K, int64_t lddy, int64_t ldx, const void *dy, const void *x) {
    static const bool off = getenv("MIVIT_NO_WGRAD_DMA") != nullptr;
    return !off && N % TILE == 0 && K % TILE == 0 && M >= 256 && lddy % 8 == 0 && ldx % 8 == 0 &&
           ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
}

size_t wgrad_dma_ws_bytes(int M, int N, int K) {
    return align_up((size_t)dma_splits(M, N, K) * N * (K + 1) * sizeof(float), 256);
}

int launch_wgrad_dma(const void *dy, int64_t lddy, const void *x, int64_t ldx, int M, int N, int K, float *dW, float *db,
                     void *ws, size_t ws_bytes, hipStream_t s) {
    MIVIT_CHECK(ws_bytes >= wgrad_dma_ws_bytes(M, N, K), "wgrad_dma: workspace too small");
    const int splits = dma_splits(M, N, K);
    int rps = ceil_div(M, splits);
    rps = (rps + BMR_MAX - 1) / BMR_MAX * BMR_MAX;
    const int nz = ceil_div(M, rps);
    float *bias_part = db ? static_cast<float *>(ws) + (size_t)nz * N * K : nullptr;
    static const int remap = getenv("MIVIT_XCD_REMAP") ? atoi(getenv("MIVIT_XCD_REMAP")) : 1;
    WgArgs a = {static_cast<const bf16 *>(dy), lddy, static_cast<const bf16 *>(x), ldx, static_cast<float *>(ws), M, N, K, rps,
                bias_part, remap};
    // occupancy beats ring depth on this part, narrow (HBM-bound) and wide (MFMA-bound) layers alike: measured on the
    // headline shape 3 slots x 64 rows (96 KB, one workgroup = 4 waves per CU) 2.35 ms/step, 2 slots (64 KB, two
    // workgroups) 1.68 ms.  MIVIT_WGRAD_DMA_CFG = 10 * slots + rows / 32 selects another point.
    // wide (MFMA-bound) layers prefer 32-row stages (32 KB ring, more workgroups per CU): c4 step 21.15 -> 20.2 ms.
    static const int forced = getenv("MIVIT_WGRAD_DMA_CFG") ? atoi(getenv("MIVIT_WGRAD_DMA_CFG")) : 0;
    const int cfg = forced ? forced : ((long)N * K >= 512L * 512L ? 21 : 22);
    size_t bytes;
    void (*kern)(const WgArgs);
    switch (cfg) {
        case 32: kern = wgrad_dma_kernel<3, 64>; bytes = 3 * 2 * 64 * TILE * 2; break;
        case 21: kern = wgrad_dma_kernel<2, 32>; bytes = 2 * 2 * 32 * TILE * 2; break;
        case 31: kern = wgrad_dma_kernel<3, 32>; bytes = 3 * 2 * 32 * TILE * 2; break;
        default: kern = wgrad_dma_kernel<2, 64>; bytes = 2 * 2 * 64 * TILE * 2; break;
    }
    MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    {
        ProfScope prof(s);
        hipLaunchKernelGGL(kern, dim3(K / TILE, N / TILE, nz), dim3(256), bytes, s, a);
        MIVIT_LAUNCH_CHECK();
    }
    int rc = launch_slab_reduce(static_cast<const float *>(ws), nz, (int64_t)N * K, dW, 0, s);
    if (rc || !db) return rc;
    return launch_slab_reduce(bias_part, nz, N, db, 0, s);
}

#ifndef MIVIT_ELEM_F16      // operator-level C-ABI: declared for bf16 (include/mivit_hip.h)
extern "C" size_t mivit_wgrad_bf16_workspace_bytes(int M, int N, int K) {
    return (N % TILE == 0 && K % TILE == 0) ? wgrad_dma_ws_bytes(M, N, K) : 0;
}
extern "C" int mivit_wgrad_bf16(const void *dy, int64_t lddy, const void *x, int64_t ldx, int M, int N, int K, float *dW,
                                float *db, void *workspace, size_t workspace_bytes, void *stream) {
    MIVIT_CHECK(dy && x && dW && workspace, "wgrad_bf16: null pointer");
    if (!wgrad_dma_supported(M, N, K, lddy, ldx, dy, x)) { mivit_set_error("wgrad_bf16: unsupported shape"); return 3; }
    prof_set_tag(MIVIT_PROF_OP);
    return launch_wgrad_dma(dy, lddy, x, ldx, M, N, K, dW, db, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}
#endif
