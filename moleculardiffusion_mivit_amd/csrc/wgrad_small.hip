// Weight gradients of NARROW layers (64-wide models: N, K in {64, 96, 128}), bf16:  dW[N, K] = dY[M, N]^T . X[M, K].
// The whole [NB x KB] gradient block (<= 32 MFMA tiles) lives in one wave's accumulators; every wave walks its own
// 32-row chunks of dY and X: 16-byte global loads one chunk ahead -> wave-private natural [row][col] LDS images ->
// both MFMA operands by ds_read_b64_tr_b16 (the contraction runs over rows).  No barrier while streaming; the eight
// waves of a workgroup are summed through LDS in a fixed order at the end, workgroups through a slab + slab_reduce
// (deterministic).  Wide layers use wgrad_dma.hip (128 x 128 tiles).
//
// XF32 (round 3): the same kernel as the weight gradient of the LINEAR FRAME EMBEDDING for small frames (reference
// helpers/models.py:146-164, patch sizes 9 x 9 = 81 ... 16 x 16 = 256 pixels; wider frames use embed.hip): X is the fp32 frame
// matrix [M, K] with ANY K -- rows are only 4-byte aligned --, read through a raw buffer (16-byte loads at dword alignment; the
// range check returns zero past the end of the tensor, so rows past M need no branch), converted to the 16-bit element type on the
// way into the LDS image; the contraction width KB is K rounded up to a multiple of 16 and the columns past K are zeroed in
// registers (they hold the next row's pixels).  Before this the embedding's weight gradient at these sizes ran on the general
// register-staged GEMM with fp32 A: 103 us of the 1.55 ms Framerate-shape step for 99 MB of frames.
#include "common.h"
#include "stream_prims.h"
#include <stdlib.h>
#include <algorithm>
#include "elem.h"          // element type of this translation unit (bf16, or f16 under -DMIVIT_ELEM_F16): after every other include

namespace {

constexpr int NWV = 8, NT = NWV * 64, CH = 32;        // waves per workgroup, rows per chunk (one 32-deep MFMA k step)


struct WsmArgs {
    const bf16 *dY; int64_t lddy; const bf16 *X; int64_t ldx;          // XF32: X points at fp32 rows, ldx in floats (= K)
    float *slabs;        // [gridDim.x][N][K]
    float *bias_slabs;   // optional [gridDim.x][N]: column sums of dY (bias gradient), one extra MFMA per dY fragment
    int M, N, K;
};

template <int W> constexpr int img_ld() { return W + (W % 128 == 0 ? 16 : 8); }   // row stride (elements): conflict-free tr reads

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// NB = dY columns of this workgroup (blockIdx.y selects the window), KB = all X columns (XF32: a.K rounded up to 16)
// NW waves per workgroup: 8, or 4 (one per SIMD, 512 registers) for the wide fp32-frame shapes whose raw chunk in flight does not
// fit beside the accumulators at 256
template <int NB, int KB, bool XF32 = false, int NW = NWV>
__global__ __launch_bounds__(NW * 64) void wgrad_small_kernel(const WsmArgs a) {
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TMN = NB / 16, TKN = KB / 16, LDA = img_ld<NB>(), LDB = img_ld<KB>();
    constexpr int VA = CH * NB / 8 / 64, VB = CH * KB / 8 / 64;          // 16-byte vectors per lane per chunk
    static_assert(TMN * TKN <= 32, "accumulator budget");
    static_assert(CH * NB / 8 % 64 == 0 && CH * KB / 8 % 64 == 0, "chunk must split evenly over the lanes");
    constexpr int WAVE_LDS = (CH * LDA + CH * LDB) * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cq = lane & 15, q = cq >> 2, p = cq & 3;
    const int n0 = blockIdx.y * NB;
    bf16 *Aimg = reinterpret_cast<bf16 *>(smem + wave * WAVE_LDS), *Bimg = Aimg + CH * LDA;

    f32x4 acc[TMN][TKN], accb[TMN];
#pragma unroll
    for (int i = 0; i < TMN; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TKN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = a.bias_slabs != nullptr;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    const int nchunks = (a.M + CH - 1) / CH, stride = gridDim.x * NW;
    uint4 ra[VA], rb[XF32 ? 2 * VB : VB];
    // XF32: the frames as one raw buffer (the launcher checks M * K * 4 < 2^32)
    const __amdgpu_buffer_rsrc_t xrows = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(a.X), 0, XF32 ? (int)((unsigned)a.M * (unsigned)a.K * 4u) : 0, 0x00020000);
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int v = 0; v < VA; ++v) {
            const int idx = v * 64 + lane, r = idx / (NB / 8), col = (idx % (NB / 8)) * 8, row = c * CH + r;
            ra[v] = row < a.M ? *reinterpret_cast<const uint4 *>(a.dY + (int64_t)row * a.lddy + n0 + col) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int v = 0; v < VB; ++v) {
            const int idx = v * 64 + lane, r = idx / (KB / 8), col = (idx % (KB / 8)) * 8, row = c * CH + r;
            if constexpr (XF32) {
                // (rows past M: at or past the end of the buffer -> zeros; < 2^32 by the launcher's check, unsigned arithmetic:
                //  a row index past M can wrap only for M * K * 4 within 32 rows of 2^32 -- excluded there)
                const int off = (int)(((unsigned)row * (unsigned)a.K + (unsigned)col) * 4u);
                const u32x4_t lo = __builtin_amdgcn_raw_buffer_load_b128(xrows, off, 0, 0), hi = __builtin_amdgcn_raw_buffer_load_b128(xrows, off + 16, 0, 0);
                rb[2 * v] = make_uint4(lo[0], lo[1], lo[2], lo[3]); rb[2 * v + 1] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            } else {
                rb[v] = row < a.M ? *reinterpret_cast<const uint4 *>(a.X + (int64_t)row * a.ldx + col) : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    int chunk = blockIdx.x * NW + wave;
    if (chunk < nchunks) load_chunk(chunk);
    for (; chunk < nchunks; chunk += stride) {
        // registers (this chunk) -> wave-private LDS images; then prefetch the next chunk while the MFMAs run
#pragma unroll
        for (int v = 0; v < VA; ++v) {
            const int idx = v * 64 + lane, r = idx / (NB / 8), col = (idx % (NB / 8)) * 8;
            *reinterpret_cast<uint4 *>(Aimg + r * LDA + col) = ra[v];
        }
#pragma unroll
        for (int v = 0; v < VB; ++v) {
            const int idx = v * 64 + lane, r = idx / (KB / 8), col = (idx % (KB / 8)) * 8;
            if constexpr (XF32) {          // fp32 pixels -> element type; columns past K (the next row's pixels) are zeroed
                const uint32_t w[8] = {rb[2 * v].x, rb[2 * v].y, rb[2 * v].z, rb[2 * v].w, rb[2 * v + 1].x, rb[2 * v + 1].y, rb[2 * v + 1].z, rb[2 * v + 1].w};
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = col + e < a.K ? __uint_as_float(w[e]) : 0.f;
                store16(Bimg + r * LDB + col, f);
            } else {
                *reinterpret_cast<uint4 *>(Bimg + r * LDB + col) = rb[v];
            }
        }
        wave_lds_fence();
        if (chunk + stride < nchunks) load_chunk(chunk + stride);
        const int rlo = 4 * g + q, rhi = rlo + 16;
        bf16x8 bf[TKN];
#pragma unroll
        for (int j = 0; j < TKN; ++j) bf[j] = tr_pair(Bimg + rlo * LDB + j * 16 + 4 * p, Bimg + rhi * LDB + j * 16 + 4 * p);
#pragma unroll
        for (int i = 0; i < TMN; ++i) {
            const bf16x8 af = tr_pair(Aimg + rlo * LDA + i * 16 + 4 * p, Aimg + rhi * LDA + i * 16 + 4 * p);
#pragma unroll
            for (int j = 0; j < TKN; ++j) acc[i][j] = mma(af, bf[j], acc[i][j]);
            if (do_bias) accb[i] = mma(af, ones, accb[i]);          // every column of the tile = sum over rows of dY[:, n]
        }
        wave_lds_fence();         // the images are overwritten at the top of the next iteration
    }
    // ---- workgroup reduction in a fixed order: waves 0-3 store into four regions, waves 4-7 add into them, then every
    //      thread sums the four regions element-wise into the workgroup's slab row (three barriers in all) ----
    __syncthreads();
    constexpr int RSZ = NB * KB + NB;                       // one region: [NB][KB] gradient block + [NB] bias sums
    float *red = reinterpret_cast<float *>(smem) + (wave & 3) * RSZ;
    for (int half = 0; half < NW / 4; ++half) {
        if ((wave >> 2) == half) {
#pragma unroll
            for (int i = 0; i < TMN; ++i) {
                float old[TKN][4];
#pragma unroll
                for (int j = 0; j < TKN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) old[j][r] = half == 0 ? 0.f : red[(i * 16 + 4 * g + r) * KB + j * 16 + cq];
#pragma unroll
                for (int j = 0; j < TKN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(i * 16 + 4 * g + r) * KB + j * 16 + cq] = old[j][r] + acc[i][j][r];
                if (do_bias && cq == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float *dst = red + NB * KB + i * 16 + 4 * g + r;
                        *dst = half == 0 ? accb[i][r] : *dst + accb[i][r];
                    }
                }
            }
        }
        __syncthreads();
    }
    const float *r0 = reinterpret_cast<const float *>(smem);
    float *out = a.slabs + (int64_t)blockIdx.x * a.N * a.K;
    for (int i = tid; i < NB * KB; i += NT) {
        const int n = i / KB, k = i - n * KB;
        if (!XF32 || k < a.K) out[(int64_t)(n0 + n) * a.K + k] = ((r0[i] + r0[RSZ + i]) + r0[2 * RSZ + i]) + r0[3 * RSZ + i];
    }
    if (do_bias)
        for (int i = tid; i < NB; i += NT) {
            const int o = NB * KB + i;
            a.bias_slabs[(int64_t)blockIdx.x * a.N + n0 + i] = ((r0[o] + r0[RSZ + o]) + r0[2 * RSZ + o]) + r0[3 * RSZ + o];
        }
}

constexpr int SLAB_PARTS = 256;           // one resident workgroup per CU

template <int NB, int KB, bool XF32 = false, int NW = NWV>
int launch_t(const WsmArgs &a, int nsplit, int *parts, hipStream_t s) {
    constexpr int NT = NW * 64;
    constexpr int WAVE_LDS = (CH * img_ld<NB>() + CH * img_ld<KB>()) * 2;
    const size_t bytes = std::max((size_t)NW * WAVE_LDS, (size_t)4 * (NB * KB + NB) * 4);
    auto kern = wgrad_small_kernel<NB, KB, XF32, NW>;
    MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    const int nchunks = ceil_div(a.M, CH);
    // short problems (<= 1024 rows): one workgroup per column window writes the result in place, no slab + reduce launches
    const int gx = nchunks <= 4 * NWV ? 1 : std::max(1, std::min(SLAB_PARTS / nsplit, ceil_div(nchunks, NW)));
    ProfScope prof(s);
    hipLaunchKernelGGL(kern, dim3(gx, nsplit), dim3(NT), bytes, s, a);
    MIVIT_LAUNCH_CHECK();
    *parts = gx;
    return 0;
}

// (N window, K) shapes instantiated: the 64-wide models' projections
int window_of(int N, int K) {
    if (K == 64 && (N == 64 || N == 128)) return N;
    if (K == 64 && N == 192) return 96;
    if (K == 128 && N == 64) return 64;
    return 0;
}

// fp32 frame rows (XF32): contraction width KB = K rounded up, dY window NB such that NB / 16 * KB / 16 <= 24 accumulator tiles
// (the raw fp32 chunk in flight takes the registers the bf16 form spends on its last 8 tiles)
int frame_kb(int K) { return K <= 64 ? 64 : K <= 96 ? 96 : K <= 128 ? 128 : K <= 192 ? 192 : K <= 256 ? 256 : 0; }
int frame_nb(int N, int K) {
    const int kb = frame_kb(K);
    if (!kb || (N != 64 && N != 128)) return 0;
    return kb <= 96 ? 64 : 32;
}

}  // namespace

bool embed_small_wgrad_supported(int M, int N, int K, int64_t lddy, const void *dy, const void *x) {
    static const bool off = getenv("MIVIT_NO_EMBED_SMALL") != nullptr;
    return !off && frame_nb(N, K) != 0 && M >= 256 && ((int64_t)M + 2 * CH) * K * 4 < (1ll << 32) && lddy % 8 == 0 &&
           (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 3) == 0;
}
size_t embed_small_wgrad_ws_bytes(int M, int N, int K) {
    return frame_nb(N, K) ? align_up((size_t)SLAB_PARTS * N * (K + 1) * sizeof(float), 256) : 0;
}
// dW [N, K] = dY^T X with X the fp32 frames [M, K] (any K <= 256), db [N] optional: same slabs + fixed-order reduction
int launch_embed_small_wgrad(const void *dy, int64_t lddy, const float *x, int M, int N, int K, float *dW, float *db, void *ws,
                             size_t ws_bytes, hipStream_t s) {
    MIVIT_CHECK(ws_bytes >= embed_small_wgrad_ws_bytes(M, N, K), "embed_small_wgrad: workspace too small");
    const bool in_place = ceil_div(M, CH) <= 4 * NWV;
    float *bias_slabs = db ? (in_place ? db : static_cast<float *>(ws) + (size_t)SLAB_PARTS * N * K) : nullptr;
    WsmArgs a = {static_cast<const bf16 *>(dy), lddy, reinterpret_cast<const bf16 *>(x), K, in_place ? dW : static_cast<float *>(ws),
                 bias_slabs, M, N, K};
    const int nb = frame_nb(N, K), kb = frame_kb(K), nsplit = N / nb;
    int parts = 0, rc;
    if (kb == 64) rc = launch_t<64, 64, true>(a, nsplit, &parts, s);
    else if (kb == 96) rc = launch_t<64, 96, true>(a, nsplit, &parts, s);
    else if (kb == 128) rc = launch_t<32, 128, true>(a, nsplit, &parts, s);
    else if (kb == 192) rc = launch_t<32, 192, true, 4>(a, nsplit, &parts, s);
    else if (kb == 256) rc = launch_t<32, 256, true, 4>(a, nsplit, &parts, s);
    else MIVIT_FAIL("embed_small_wgrad: unsupported shape N=%d K=%d", N, K);
    if (rc) return rc;
    if (in_place) return 0;
    if ((rc = launch_slab_reduce(static_cast<const float *>(ws), parts, (int64_t)N * K, dW, 0, s))) return rc;
    return db ? launch_slab_reduce(bias_slabs, parts, N, db, 0, s) : 0;
}

bool wgrad_small_supported(int M, int N, int K, int64_t lddy, int64_t ldx, const void *dy, const void *x) {
    static const bool off = getenv("MIVIT_NO_WGRAD_SMALL") != nullptr;
    return !off && window_of(N, K) != 0 && M >= 256 && lddy % 8 == 0 && ldx % 8 == 0 &&
           ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
}

size_t wgrad_small_ws_bytes(int M, int N, int K) {
    return window_of(N, K) ? align_up((size_t)SLAB_PARTS * N * (K + 1) * sizeof(float), 256) : 0;
}

int launch_wgrad_small(const void *dy, int64_t lddy, const void *x, int64_t ldx, int M, int N, int K, float *dW, float *db,
                       void *ws, size_t ws_bytes, hipStream_t s) {
    MIVIT_CHECK(ws_bytes >= wgrad_small_ws_bytes(M, N, K), "wgrad_small: workspace too small");
    const bool in_place = ceil_div(M, CH) <= 4 * NWV;          // must match launch_t's single-workgroup rule
    float *bias_slabs = db ? (in_place ? db : static_cast<float *>(ws) + (size_t)SLAB_PARTS * N * K) : nullptr;
    WsmArgs a = {static_cast<const bf16 *>(dy), lddy, static_cast<const bf16 *>(x), ldx, in_place ? dW : static_cast<float *>(ws),
                 bias_slabs, M, N, K};
    const int nb = window_of(N, K), nsplit = N / nb;
    int parts = 0, rc;
    if (nb == 64 && K == 64) rc = launch_t<64, 64>(a, nsplit, &parts, s);
    else if (nb == 128 && K == 64) rc = launch_t<128, 64>(a, nsplit, &parts, s);
    else if (nb == 96 && K == 64) rc = launch_t<96, 64>(a, nsplit, &parts, s);
    else if (nb == 64 && K == 128) rc = launch_t<64, 128>(a, nsplit, &parts, s);
    else MIVIT_FAIL("wgrad_small: unsupported shape N=%d K=%d", N, K);
    if (rc) return rc;
    if (in_place) return 0;
    if ((rc = launch_slab_reduce(static_cast<const float *>(ws), parts, (int64_t)N * K, dW, 0, s))) return rc;
    return db ? launch_slab_reduce(bias_slabs, parts, N, db, 0, s) : 0;
}

#ifndef MIVIT_ELEM_F16      // operator-level C-ABI: declared for bf16 (include/mivit_hip.h)
extern "C" size_t mivit_embed_small_wgrad_workspace_bytes(int M, int K, int E) { return embed_small_wgrad_ws_bytes(M, E, K); }
extern "C" int mivit_embed_small_wgrad(const void *dY_bf16, const float *X, int M, int K, int E, float *dW, float *db, void *workspace,
                                       size_t workspace_bytes, void *stream) {
    MIVIT_CHECK(dY_bf16 && X && dW && workspace, "embed_small_wgrad: null pointer");
    if (!embed_small_wgrad_supported(M, E, K, E, dY_bf16, X)) { mivit_set_error("embed_small_wgrad: unsupported shape"); return 3; }
    prof_set_tag(MIVIT_PROF_OP);
    return launch_embed_small_wgrad(dY_bf16, E, X, M, E, K, dW, db, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}
extern "C" size_t mivit_wgrad_small_workspace_bytes(int M, int N, int K) {
    return window_of(N, K) ? wgrad_small_ws_bytes(M, N, K) : 0;
}
extern "C" int mivit_wgrad_small(const void *dy, int64_t lddy, const void *x, int64_t ldx, int M, int N, int K, float *dW,
                                 float *db, void *workspace, size_t workspace_bytes, void *stream) {
    MIVIT_CHECK(dy && x && dW && workspace, "wgrad_small: null pointer");
    if (!wgrad_small_supported(M, N, K, lddy, ldx, dy, x)) { mivit_set_error("wgrad_small: unsupported shape"); return 3; }
    prof_set_tag(MIVIT_PROF_OP);
    return launch_wgrad_small(dy, lddy, x, ldx, M, N, K, dW, db, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}
#endif
