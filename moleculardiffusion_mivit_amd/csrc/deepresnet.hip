// DeepResNetEmbedding, inference (eval-mode BatchNorm) -- reference helpers/models.py:230-257 + ResidualBlock :202-228.
//
// A frame is tiny (9x9 .. 13x13 pixels), so ONE workgroup carries F whole frames through the entire conv stack
// without touching HBM in between: activations live in two LDS buffers as zero-haloed [pixel][channel] images, every
// 3x3 / 1x1 convolution is an implicit GEMM on MFMA (rows = output pixels, k = (tap, input channel): the A fragment of
// a k-step is 8 contiguous channels of one neighbour pixel -> one LDS read; B fragments are read straight from the
// L2-resident weights), global average pooling comes from the accumulators, the final Linear runs on the VALU.
// Eval-mode BatchNorm is an affine map per channel: the host folds the scale into the conv weights (laid out
// [c_out][tap][c_in]) and the shifts into one bias per conv, so a residual block is
//      t   = relu(conv3x3'(x)  + b_a)
//      out = relu(conv3x3'(t) + conv1x1'(x) + b_b)          (both convs accumulate into the same MFMA tile)
// HBM traffic per frame: P*P floats in, E floats out.
// Training (batch-statistics BatchNorm needs a grid-wide reduction per layer) still runs on PyTorch-ROCm.
#include "common.h"

namespace {

struct DrnArgs {
    const float *x;                       // [N, P, P]
    int N, P, E, F;                       // F = frames per block
    const float *w0, *b0;                 // conv0: [32][9], [32]               (fp32, VALU)
    const void *w11, *w12, *w1s;          // block 1: [64][9][32], [64][9][64], [64][1][32]    (T)
    const void *w21, *w22, *w2s;          // block 2: [128][9][64], [128][9][128], [128][1][64]
    const float *b11, *b12, *b21, *b22;   // folded biases
    const float *wfc, *bfc;               // [E][128], [E]
    float *tokens;                        // [N, E]
};

constexpr int NT = 512, NWAVE = 8, MAXM = 11;     // <= 176 output pixels per block
constexpr int PADB = 16;                          // bytes of padding per pixel: neighbouring pixels land in different banks
template <typename T> constexpr int pad_el() { return PADB / (int)sizeof(T); }

template <typename T>
__device__ __forceinline__ typename Mma<T>::Frag frag_at(const T *p);
template <>
__device__ __forceinline__ float frag_at<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ bf16x8 frag_at<bf16>(const bf16 *p) { return *reinterpret_cast<const bf16x8 *>(p); }

// acc[mt] += sum over taps, input channels of  in[pixel(mt) + tap][ci] * W[co][tap][ci]   for this wave's column tile
template <typename T, int CIN, int TAPS>
__device__ __forceinline__ void conv_accum(f32x4 (&acc)[MAXM], const T *in, const T *W, const int (&hidx)[MAXM], int mt0,
                                           int mt1, int nt, int HW2, int lane) {
    constexpr int KS = Mma<T>::KS;
    constexpr int KL = (sizeof(T) == 2) ? 8 : 1;            // k elements per lane-group step
    const int g = lane >> 4, cq = lane & 15;
    constexpr int CS = CIN + pad_el<T>();                   // pixel stride in LDS
    const T *wrow = W + (size_t)(nt * 16 + cq) * TAPS * CIN + g * KL;
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
        const int off = TAPS == 9 ? ((tap / 3 - 1) * HW2 + (tap % 3 - 1)) : 0;
#pragma unroll
        for (int c0 = 0; c0 < CIN; c0 += KS) {
            const typename Mma<T>::Frag b = frag_at<T>(wrow + tap * CIN + c0);
#pragma unroll
            for (int mt = 0; mt < MAXM; ++mt)
                if (mt >= mt0 && mt < mt1)
                    acc[mt] = Mma<T>::mma(frag_at<T>(in + (hidx[mt] + off) * CS + c0 + g * KL), b, acc[mt]);
        }
    }
}

template <typename T>
__device__ __forceinline__ void zero_halo(T *buf, int C, int F, int P, int tid) {
    const int HW2 = P + 2, HP = HW2 * HW2, border = 4 * (P + 1);
    for (int i = tid; i < F * border * C; i += NT) {
        const int c = i % C, cell = (i / C) % border, f = i / (C * border);
        int y, x;
        if (cell < HW2) { y = 0; x = cell; }
        else if (cell < 2 * HW2) { y = HW2 - 1; x = cell - HW2; }
        else if (cell < 2 * HW2 + P) { y = cell - 2 * HW2 + 1; x = 0; }
        else { y = cell - 2 * HW2 - P + 1; x = HW2 - 1; }
        buf[(f * HP + y * HW2 + x) * (C + pad_el<T>()) + c] = from_f32<T>(0.f);
    }
}

// halo index of logical output row r (r = f * P*P + y * P + x); rows past the last frame alias the first interior pixel
// (their results are never stored; every neighbour of an interior pixel is inside the buffer)
__device__ __forceinline__ int halo_index(int r, int rows, int P) {
    if (r >= rows) return P + 3;
    const int pp = P * P, f = r / pp, rem = r - f * pp, y = rem / P, x = rem - y * P;
    return f * (P + 2) * (P + 2) + (y + 1) * (P + 2) + x + 1;
}

template <typename T, int COUT>
__device__ __forceinline__ void store_act(const f32x4 (&acc)[MAXM], T *out, const float *bias, int mt0, int mt1, int nt,
                                          int rows, int P, int lane) {
    const int g = lane >> 4, co = nt * 16 + (lane & 15);
    const float bv = bias[co];
#pragma unroll
    for (int mt = 0; mt < MAXM; ++mt)
        if (mt >= mt0 && mt < mt1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = mt * 16 + 4 * g + r;
                if (row < rows) out[halo_index(row, rows, P) * (COUT + pad_el<T>()) + co] = from_f32<T>(fmaxf(acc[mt][r] + bv, 0.f));
            }
        }
}

template <typename T>
__global__ __launch_bounds__(NT) void deepresnet_eval_kernel(const DrnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P, HW2 = P + 2, HP = HW2 * HW2, PP = P * P;
    const int f0 = blockIdx.x * a.F;
    const int nf = min(a.F, a.N - f0);              // frames of this block
    const int rows = nf * PP, NM = (rows + 15) / 16;
    T *bufA = reinterpret_cast<T *>(smem);
    const int BUF = a.F * HP * (128 + pad_el<T>());
    T *bufB = bufA + BUF;
    float *pool = reinterpret_cast<float *>(bufB + BUF);   // [F][128]

    // everything zero once (halos of the first layouts, rows of absent frames)
    for (int i = tid; i < (int)(2 * BUF * sizeof(T) / 4); i += NT) reinterpret_cast<uint32_t *>(smem)[i] = 0u;
    __syncthreads();

    // ---- conv0 (1 -> 32) + folded BN + ReLU on the VALU: bufA as [pixel][32] ----
    for (int i = tid; i < rows * 32; i += NT) {
        const int co = i & 31, r = i >> 5;
        const int f = r / PP, rem = r - f * PP, y = rem / P, x = rem - y * P;
        const float *img = a.x + (size_t)(f0 + f) * PP;
        float s = a.b0[co];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (yy >= 0 && yy < P && xx >= 0 && xx < P) s += img[yy * P + xx] * a.w0[co * 9 + t];
        }
        bufA[halo_index(r, rows, P) * (32 + pad_el<T>()) + co] = from_f32<T>(fmaxf(s, 0.f));
    }
    __syncthreads();

    int hidx[MAXM];
#pragma unroll
    for (int mt = 0; mt < MAXM; ++mt) hidx[mt] = halo_index(mt * 16 + (lane & 15), rows, P);
    f32x4 acc[MAXM];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    // ================= residual block 1: 32 -> 64 channels (4 column tiles: 2 waves share a tile, splitting rows) =========
    {
        const int nt = wave & 3, half = wave >> 2, hm = (NM + 1) / 2;
        const int mt0 = half * hm, mt1 = min(NM, mt0 + hm);
#pragma unroll
        for (int mt = 0; mt < MAXM; ++mt) acc[mt] = zero;
        conv_accum<T, 32, 9>(acc, bufA, static_cast<const T *>(a.w11), hidx, mt0, mt1, nt, HW2, lane);
        store_act<T, 64>(acc, bufB, a.b11, mt0, mt1, nt, rows, P, lane);          // t = relu(conv1' + b): bufB [pixel][64]
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MAXM; ++mt) acc[mt] = zero;
        conv_accum<T, 64, 9>(acc, bufB, static_cast<const T *>(a.w12), hidx, mt0, mt1, nt, HW2, lane);
        conv_accum<T, 32, 1>(acc, bufA, static_cast<const T *>(a.w1s), hidx, mt0, mt1, nt, HW2, lane);
        __syncthreads();                                                           // bufA (x) fully consumed
        zero_halo<T>(bufA, 64, a.F, P, tid);
        store_act<T, 64>(acc, bufA, a.b12, mt0, mt1, nt, rows, P, lane);          // out1: bufA [pixel][64]
        zero_halo<T>(bufB, 128, a.F, P, tid);                                      // bufB (t) fully consumed too
        __syncthreads();
    }
    // ================= residual block 2: 64 -> 128 channels (8 column tiles: one per wave, all rows) =======================
    {
        const int nt = wave;
#pragma unroll
        for (int mt = 0; mt < MAXM; ++mt) acc[mt] = zero;
        conv_accum<T, 64, 9>(acc, bufA, static_cast<const T *>(a.w21), hidx, 0, NM, nt, HW2, lane);
        store_act<T, 128>(acc, bufB, a.b21, 0, NM, nt, rows, P, lane);            // t': bufB [pixel][128]
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MAXM; ++mt) acc[mt] = zero;
        conv_accum<T, 128, 9>(acc, bufB, static_cast<const T *>(a.w22), hidx, 0, NM, nt, HW2, lane);
        conv_accum<T, 64, 1>(acc, bufA, static_cast<const T *>(a.w2s), hidx, 0, NM, nt, HW2, lane);
        // ---- global average pool of relu(acc + b) straight from the accumulators (this wave owns its 16 channels) ----
        const int g = lane >> 4, co = nt * 16 + (lane & 15);
        const float bv = a.b22[co];
        for (int f = 0; f < nf; ++f) {
            float s = 0.f;
#pragma unroll
            for (int mt = 0; mt < MAXM; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mt * 16 + 4 * g + r;
                    if (mt < NM && row >= f * PP && row < (f + 1) * PP) s += fmaxf(acc[mt][r] + bv, 0.f);
                }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (g == 0) pool[f * 128 + co] = s / (float)PP;
        }
        __syncthreads();
    }
    // ================= fc: tokens[f][e] = pool[f] . wfc[e] + bfc[e] =========================================================
    for (int i = tid; i < nf * a.E; i += NT) {
        const int e = i % a.E, f = i / a.E;
        const float *w = a.wfc + (size_t)e * 128, *pl = pool + f * 128;
        float s = a.bfc[e];
#pragma unroll 8
        for (int c = 0; c < 128; ++c) s += pl[c] * w[c];
        a.tokens[(size_t)(f0 + f) * a.E + e] = s;
    }
}

size_t drn_lds(int dtype, int P, int F) {
    return (size_t)2 * F * (P + 2) * (P + 2) * (128 * dtype_size(dtype) + PADB) + (size_t)F * 128 * 4;
}

int drn_frames_per_block(int dtype, int P) {
    int F = (MAXM * 16) / (P * P);
    while (F >= 1 && drn_lds(dtype, P, F) > 160 * 1024) --F;
    return F;
}

}  // namespace

extern "C" int mivit_deepresnet_eval_supported(int dtype, int patch_size) {
    return patch_size >= 3 && drn_frames_per_block(dtype, patch_size) >= 1;
}

extern "C" int mivit_deepresnet_eval_fwd(int dtype, const float *x, int N, int P, int E, const float *w0, const float *b0,
                                         const void *w11, const void *w12, const void *w1s, const void *w21,
                                         const void *w22, const void *w2s, const float *b11, const float *b12,
                                         const float *b21, const float *b22, const float *wfc, const float *bfc,
                                         float *tokens, void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16, "bad dtype %d", dtype);
    MIVIT_CHECK(x && w0 && b0 && w11 && w12 && w1s && w21 && w22 && w2s && b11 && b12 && b21 && b22 && wfc && bfc && tokens,
                "deepresnet_eval_fwd: null pointer");
    MIVIT_CHECK(N > 0 && E > 0, "deepresnet_eval_fwd: empty problem");
    const int F = drn_frames_per_block(dtype, P);
    if (P < 3 || F < 1) { mivit_set_error("deepresnet_eval_fwd: frame side %d does not fit the LDS-resident kernel", P); return 3; }
    DrnArgs a = {x, N, P, E, F, w0, b0, w11, w12, w1s, w21, w22, w2s, b11, b12, b21, b22, wfc, bfc, tokens};
    const size_t lds = drn_lds(dtype, P, F);
    hipStream_t s = static_cast<hipStream_t>(stream);
    prof_set_tag(MIVIT_PROF_OP);
    ProfScope prof(s);
    if (dtype == MIVIT_F32) {
        MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(deepresnet_eval_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(deepresnet_eval_kernel<float>, dim3(ceil_div(N, F)), dim3(NT), lds, s, a);
    } else {
        MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(deepresnet_eval_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(deepresnet_eval_kernel<bf16>, dim3(ceil_div(N, F)), dim3(NT), lds, s, a);
    }
    MIVIT_LAUNCH_CHECK();
    return 0;
}
