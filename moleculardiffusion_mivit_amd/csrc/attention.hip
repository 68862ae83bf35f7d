// Multi-head self-attention core, forward and backward (reference helpers/models.py:42-54 and its autograd).
//
// Sequences are short (S = frames + 1 <= ~128), so one 64-lane wavefront owns one (batch, head) pair and keeps
// Q, K, V (and dO in backward) of the whole sequence in LDS: scores, softmax and P.V never touch HBM and the
// backward recomputes the probabilities instead of saving them.  All five products run on MFMA 16x16 tiles
// (v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32); softmax rows are reduced with 16-lane shuffles in the
// accumulator layout; statistics are fp32.
//
// LDS images are [row][k] with k contiguous.  bf16 fragments need 8 contiguous k, so operands that are consumed
// along their other axis get an explicit transposed copy; the fp32 fragment is a single dword, so the fp32 build
// reads the natural image with swapped strides instead (no copies).
#include "common.h"

namespace {

template <typename T>
struct Img {
    T *p;
    int rs, ks;
};

template <typename T>
__device__ __forceinline__ f32x4 tile_mma(const Img<T> A, int ar0, const Img<T> B, int br0, int kdim, int lane) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < kdim; kk += Mma<T>::KS)
        acc = Mma<T>::mma(Mma<T>::load(A.p, A.rs, A.ks, ar0, kk, lane), Mma<T>::load(B.p, B.rs, B.ks, br0, kk, lane), acc);
    return acc;
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct AttnDims {
    int S, Dh, E, H, B;
    int Sp;     // S rounded up to 16 (tile rows / key columns)
    int Skp;    // S rounded up to max(16, KS) (reduction over rows / keys)
    int Dp;     // Dh rounded up to KS (reduction over d)
    int D16;    // Dh rounded up to 16 (output d tiles)
    int ldq;    // row length of [token][d] images
    int ldk;    // row length of [.][token] images
    int per_wave;   // LDS elements per wave
    int two_pass;   // fp32 backward, long sequences: ONE score image (P for dV first, then recomputed dS in place)
    int nw;     // waves per block
};

template <typename T>
__host__ __device__ inline int round_up_i(int x, int m) { return (x + m - 1) / m * m; }

template <typename T>
AttnDims make_dims(int B, int S, int H, int Dh, bool backward) {
    constexpr int KS = Mma<T>::KS;
    constexpr bool BF = sizeof(T) == 2;
    AttnDims d;
    d.B = B; d.S = S; d.H = H; d.Dh = Dh; d.E = H * Dh;
    d.Sp = (S + 15) / 16 * 16;
    const int kr = KS > 16 ? KS : 16;
    d.Skp = (S + kr - 1) / kr * kr;
    d.Dp = (Dh + KS - 1) / KS * KS;
    d.D16 = (Dh + 15) / 16 * 16;
    const int rowsq = d.Sp;                  // token rows of natural images
    if (BF) {
        d.ldq = d.Dp + 8;                    // multiple of 8 elements (16-byte fragment reads)
        d.ldk = d.Skp + 8;
    } else {
        d.ldq = (d.Dp > d.D16 ? d.Dp : d.D16) + 1;
        d.ldk = d.Skp + 1;
    }
    const int nat = rowsq * d.ldq;           // one natural [token][d] image
    const int tr = d.D16 * d.ldk;            // one transposed [d][token] image (bf16 only)
    const int ps = (d.Sp > d.Skp ? d.Sp : d.Skp) * d.ldk;   // one [token][token] image
    if (!backward) d.per_wave = 3 * nat + (BF ? tr : 0) + ps;                 // Q,K,V (+Vt) + P
    else d.per_wave = 4 * nat + (BF ? 3 * tr : 0) + (BF ? 3 : 2) * ps;        // Q,K,V,dO (+Qt,Kt,dOt) + P/dS images
    d.two_pass = 0;
    if (backward && !BF && (size_t)d.per_wave * sizeof(T) > (size_t)160 * 1024) { d.two_pass = 1; d.per_wave = 4 * nat + ps; }
    d.per_wave = (d.per_wave + 7) / 8 * 8;
    const size_t bytes = (size_t)d.per_wave * sizeof(T);
    int nw = (int)((size_t)64 * 1024 / bytes);
    d.nw = nw < 1 ? 1 : (nw > 4 ? 4 : nw);
    return d;
}

// stage one [S][Dh] head slice (row stride src_ld) into a natural image (and optionally its transpose)
template <typename T>
__device__ __forceinline__ void stage_head(const T *src, int64_t src_ld, int S, int Dh, T *nat, int ldq, T *tr, int ldk,
                                           int lane) {
    if ((Dh & 3) == 0) {
        const int per_row = Dh >> 2;
        for (int u = lane; u < S * per_row; u += 64) {
            const int r = u / per_row, d = (u % per_row) * 4;
            T v[4];
            if (sizeof(T) == 2) *reinterpret_cast<uint2 *>(v) = *reinterpret_cast<const uint2 *>(src + (int64_t)r * src_ld + d);
            else *reinterpret_cast<float4 *>(v) = *reinterpret_cast<const float4 *>(src + (int64_t)r * src_ld + d);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                nat[r * ldq + d + i] = v[i];
                if (tr) tr[(d + i) * ldk + r] = v[i];
            }
        }
    } else {
        for (int u = lane; u < S * Dh; u += 64) {
            const int r = u / Dh, d = u % Dh;
            const T v = src[(int64_t)r * src_ld + d];
            nat[r * ldq + d] = v;
            if (tr) tr[d * ldk + r] = v;
        }
    }
}

template <typename T>
__device__ __forceinline__ void zero_lds(T *p, int n, int lane) {
    uint32_t *q = reinterpret_cast<uint32_t *>(p);
    const int words = n * (int)sizeof(T) / 4;
    for (int i = lane; i < words; i += 64) q[i] = 0u;
}

// -------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------
template <typename T, int NT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T *qkv, T *ctx, const AttnDims d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr bool BF = sizeof(T) == 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * d.nw + wave;
    if (pair >= d.B * d.H) return;
    const int b = pair / d.H, h = pair % d.H;
    T *base = reinterpret_cast<T *>(smem_raw) + (size_t)wave * d.per_wave;
    const int nat = d.Sp * d.ldq;
    T *Qn = base, *Kn = Qn + nat, *Vn = Kn + nat;
    T *Vt = BF ? Vn + nat : nullptr;
    T *Ps = (BF ? Vt + d.D16 * d.ldk : Vn + nat);
    zero_lds(base, d.per_wave, lane);
    wave_lds_sync();
    const int64_t ld3 = 3 * (int64_t)d.E;
    const T *src = qkv + (int64_t)b * d.S * ld3 + h * d.Dh;
    stage_head<T>(src, ld3, d.S, d.Dh, Qn, d.ldq, nullptr, 0, lane);
    stage_head<T>(src + d.E, ld3, d.S, d.Dh, Kn, d.ldq, nullptr, 0, lane);
    stage_head<T>(src + 2 * d.E, ld3, d.S, d.Dh, Vn, d.ldq, Vt, d.ldk, lane);
    wave_lds_sync();

    const Img<T> Qi = {Qn, d.ldq, 1}, Ki = {Kn, d.ldq, 1}, Pi = {Ps, d.ldk, 1};
    const Img<T> Vti = BF ? Img<T>{Vt, d.ldk, 1} : Img<T>{Vn, 1, d.ldq};
    const float scale = 1.0f / sqrtf((float)d.Dh);
    const int colq = lane & 15, rowq = (lane >> 4) * 4;

    for (int it = 0; it < NT; ++it) {
        f32x4 sc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) sc[j] = tile_mma<T>(Qi, it * 16, Ki, j * 16, d.Dp, lane);
        // row softmax: a row's columns sit in the 16 lanes sharing (lane >> 4), across the NT key tiles
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const bool ok = j * 16 + colq < d.S;
                sc[j][r] = ok ? sc[j][r] * scale : -INFINITY;
                m = fmaxf(m, sc[j][r]);
            }
            m = group16_max(m);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float e = (j * 16 + colq < d.S) ? __expf(sc[j][r] - m) : 0.f;
                sc[j][r] = e;
                sum += e;
            }
            sum = group16_sum(sum);
            const float inv = 1.f / sum;
#pragma unroll
            for (int j = 0; j < NT; ++j) Ps[(it * 16 + rowq + r) * d.ldk + j * 16 + colq] = from_f32<T>(sc[j][r] * inv);
        }
        wave_lds_sync();
        for (int jd = 0; jd < d.D16 / 16; ++jd) {
            const f32x4 o = tile_mma<T>(Pi, it * 16, Vti, jd * 16, d.Skp, lane);
            const int dcol = jd * 16 + colq;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = it * 16 + rowq + r;
                if (row < d.S && dcol < d.Dh)
                    ctx[((int64_t)b * d.S + row) * d.E + h * d.Dh + dcol] = from_f32<T>(o[r]);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// backward: dq, dk, dv from dctx; probabilities recomputed
// -------------------------------------------------------------------------------------------------
template <typename T, int NT>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T *qkv, const T *dctx, T *dqkv, const AttnDims d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr bool BF = sizeof(T) == 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * d.nw + wave;
    if (pair >= d.B * d.H) return;
    const int b = pair / d.H, h = pair % d.H;
    T *base = reinterpret_cast<T *>(smem_raw) + (size_t)wave * d.per_wave;
    const int nat = d.Sp * d.ldq, tr = d.D16 * d.ldk;
    const int ps = (d.Sp > d.Skp ? d.Sp : d.Skp) * d.ldk;
    T *Qn = base, *Kn = Qn + nat, *Vn = Kn + nat, *On = Vn + nat;
    T *cur = On + nat;
    T *Qt = nullptr, *Kt = nullptr, *Ot = nullptr;
    if (BF) { Qt = cur; Kt = Qt + tr; Ot = Kt + tr; cur = Ot + tr; }
    T *Pa = cur;            // bf16: P^T [key][row];  f32: P [row][key]
    T *dSn = Pa + ps;       // dS   [row][key]
    T *dSt = BF ? dSn + ps : nullptr;   // dS^T [key][row] (bf16 only)
    zero_lds(base, d.per_wave, lane);
    wave_lds_sync();
    const int64_t ld3 = 3 * (int64_t)d.E;
    const T *src = qkv + (int64_t)b * d.S * ld3 + h * d.Dh;
    stage_head<T>(src, ld3, d.S, d.Dh, Qn, d.ldq, Qt, d.ldk, lane);
    stage_head<T>(src + d.E, ld3, d.S, d.Dh, Kn, d.ldq, Kt, d.ldk, lane);
    stage_head<T>(src + 2 * d.E, ld3, d.S, d.Dh, Vn, d.ldq, nullptr, 0, lane);
    stage_head<T>(dctx + (int64_t)b * d.S * d.E + h * d.Dh, d.E, d.S, d.Dh, On, d.ldq, Ot, d.ldk, lane);
    wave_lds_sync();

    const Img<T> Qi = {Qn, d.ldq, 1}, Ki = {Kn, d.ldq, 1}, Vi = {Vn, d.ldq, 1}, Oi = {On, d.ldq, 1};
    const float scale = 1.0f / sqrtf((float)d.Dh);
    const int colq = lane & 15, rowq = (lane >> 4) * 4;

    // mode 0: write P and dS (two images);  mode 1: write P only;  mode 2: write dS into the P image (two-pass fp32)
    auto score_pass = [&](const int mode) {
        for (int it = 0; it < NT; ++it) {
            f32x4 sc[NT], dp[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                sc[j] = tile_mma<T>(Qi, it * 16, Ki, j * 16, d.Dp, lane);
                dp[j] = tile_mma<T>(Oi, it * 16, Vi, j * 16, d.Dp, lane);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const bool ok = j * 16 + colq < d.S;
                    sc[j][r] = ok ? sc[j][r] * scale : -INFINITY;
                    m = fmaxf(m, sc[j][r]);
                }
                m = group16_max(m);
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float e = (j * 16 + colq < d.S) ? __expf(sc[j][r] - m) : 0.f;
                    sc[j][r] = e;
                    sum += e;
                }
                sum = group16_sum(sum);
                const float inv = 1.f / sum;
                float delta = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    sc[j][r] *= inv;                       // P
                    delta += sc[j][r] * dp[j][r];
                }
                delta = group16_sum(delta);
                const int row = it * 16 + rowq + r;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int key = j * 16 + colq;
                    const float p = sc[j][r];
                    const float ds = scale * p * (dp[j][r] - delta);
                    if (mode == 0) {
                        if (BF) {
                            Pa[key * d.ldk + row] = from_f32<T>(p);
                            dSt[key * d.ldk + row] = from_f32<T>(ds);
                        } else {
                            Pa[row * d.ldk + key] = from_f32<T>(p);
                        }
                        dSn[row * d.ldk + key] = from_f32<T>(ds);
                    } else {
                        Pa[row * d.ldk + key] = from_f32<T>(mode == 1 ? p : ds);
                    }
                }
            }
        }
    };
    const bool two_pass = !BF && d.two_pass;
    T *dst = dqkv + (int64_t)b * d.S * ld3 + h * d.Dh;
    if (two_pass) {
        // pass 1: P -> dV = P^T dO
        score_pass(1);
        wave_lds_sync();
        const Img<T> PTi1 = {Pa, 1, d.ldk}, Oti1 = {On, 1, d.ldq};
        for (int it = 0; it < NT; ++it)
            for (int jd = 0; jd < d.D16 / 16; ++jd) {
                const f32x4 dv = tile_mma<T>(PTi1, it * 16, Oti1, jd * 16, d.Skp, lane);
                const int dcol = jd * 16 + colq;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = it * 16 + rowq + r;
                    if (row < d.S && dcol < d.Dh) dst[(int64_t)row * ld3 + dcol + 2 * d.E] = from_f32<T>(dv[r]);
                }
            }
        wave_lds_sync();
        // pass 2: scores recomputed, dS written over the P image -> dQ = dS K, dK = dS^T Q
        score_pass(2);
        wave_lds_sync();
        const Img<T> dSi2 = {Pa, d.ldk, 1}, dSTi2 = {Pa, 1, d.ldk}, Qti2 = {Qn, 1, d.ldq}, Kti2 = {Kn, 1, d.ldq};
        for (int it = 0; it < NT; ++it)
            for (int jd = 0; jd < d.D16 / 16; ++jd) {
                const f32x4 dq = tile_mma<T>(dSi2, it * 16, Kti2, jd * 16, d.Skp, lane);
                const f32x4 dk = tile_mma<T>(dSTi2, it * 16, Qti2, jd * 16, d.Skp, lane);
                const int dcol = jd * 16 + colq;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = it * 16 + rowq + r;
                    if (row < d.S && dcol < d.Dh) {
                        T *o = dst + (int64_t)row * ld3 + dcol;
                        o[0] = from_f32<T>(dq[r]);
                        o[d.E] = from_f32<T>(dk[r]);
                    }
                }
            }
        return;
    }
    score_pass(0);
    wave_lds_sync();

    // operand views for the three output products
    const Img<T> PTi = BF ? Img<T>{Pa, d.ldk, 1} : Img<T>{Pa, 1, d.ldk};        // [key][row]
    const Img<T> dSTi = BF ? Img<T>{dSt, d.ldk, 1} : Img<T>{dSn, 1, d.ldk};     // [key][row]
    const Img<T> dSi = {dSn, d.ldk, 1};                                         // [row][key]
    const Img<T> Oti = BF ? Img<T>{Ot, d.ldk, 1} : Img<T>{On, 1, d.ldq};        // [d][row]
    const Img<T> Qti = BF ? Img<T>{Qt, d.ldk, 1} : Img<T>{Qn, 1, d.ldq};        // [d][row]
    const Img<T> Kti = BF ? Img<T>{Kt, d.ldk, 1} : Img<T>{Kn, 1, d.ldq};        // [d][key]
    for (int it = 0; it < NT; ++it) {
        for (int jd = 0; jd < d.D16 / 16; ++jd) {
            const f32x4 dq = tile_mma<T>(dSi, it * 16, Kti, jd * 16, d.Skp, lane);
            const f32x4 dk = tile_mma<T>(dSTi, it * 16, Qti, jd * 16, d.Skp, lane);
            const f32x4 dv = tile_mma<T>(PTi, it * 16, Oti, jd * 16, d.Skp, lane);
            const int dcol = jd * 16 + colq;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = it * 16 + rowq + r;
                if (row < d.S && dcol < d.Dh) {
                    T *o = dst + (int64_t)row * ld3 + dcol;
                    o[0] = from_f32<T>(dq[r]);
                    o[d.E] = from_f32<T>(dk[r]);
                    o[2 * d.E] = from_f32<T>(dv[r]);
                }
            }
        }
    }
}

constexpr size_t LDS_LIMIT = 160 * 1024;

template <typename T, int NT>
int launch_fwd_nt(const T *qkv, T *ctx, const AttnDims &d, hipStream_t s) {
    const size_t bytes = (size_t)d.per_wave * sizeof(T) * d.nw;
    auto kern = attn_fwd_kernel<T, NT>;
    if (bytes > 48 * 1024)
        MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    const int blocks = ceil_div(d.B * d.H, d.nw);
    ProfScope prof(s);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * d.nw), bytes, s, qkv, ctx, d);
    MIVIT_LAUNCH_CHECK();
    return 0;
}
template <typename T, int NT>
int launch_bwd_nt(const T *qkv, const T *dctx, T *dqkv, const AttnDims &d, hipStream_t s) {
    const size_t bytes = (size_t)d.per_wave * sizeof(T) * d.nw;
    auto kern = attn_bwd_kernel<T, NT>;
    if (bytes > 48 * 1024)
        MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    const int blocks = ceil_div(d.B * d.H, d.nw);
    ProfScope prof(s);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * d.nw), bytes, s, qkv, dctx, dqkv, d);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int attn_fwd_t(const void *qkv, int B, int S, int H, int Dh, void *ctx, hipStream_t s) {
    const AttnDims d = make_dims<T>(B, S, H, Dh, false);
    MIVIT_CHECK((size_t)d.per_wave * sizeof(T) <= LDS_LIMIT, "attention_fwd: S=%d Dh=%d does not fit in LDS", S, Dh);
    const T *q = static_cast<const T *>(qkv);
    T *c = static_cast<T *>(ctx);
    switch (d.Sp / 16) {
        case 1: return launch_fwd_nt<T, 1>(q, c, d, s);
        case 2: return launch_fwd_nt<T, 2>(q, c, d, s);
        case 3: return launch_fwd_nt<T, 3>(q, c, d, s);
        case 4: return launch_fwd_nt<T, 4>(q, c, d, s);
        case 5: return launch_fwd_nt<T, 5>(q, c, d, s);
        case 6: return launch_fwd_nt<T, 6>(q, c, d, s);
        case 7: return launch_fwd_nt<T, 7>(q, c, d, s);
        case 8: return launch_fwd_nt<T, 8>(q, c, d, s);
        default: MIVIT_FAIL("attention_fwd: S=%d > 128 is not supported by the LDS-resident kernel", S);
    }
}
template <typename T>
int attn_bwd_t(const void *qkv, const void *dctx, int B, int S, int H, int Dh, void *dqkv, hipStream_t s) {
    const AttnDims d = make_dims<T>(B, S, H, Dh, true);
    MIVIT_CHECK((size_t)d.per_wave * sizeof(T) <= LDS_LIMIT, "attention_bwd: S=%d Dh=%d does not fit in LDS", S, Dh);
    const T *q = static_cast<const T *>(qkv);
    const T *g = static_cast<const T *>(dctx);
    T *o = static_cast<T *>(dqkv);
    switch (d.Sp / 16) {
        case 1: return launch_bwd_nt<T, 1>(q, g, o, d, s);
        case 2: return launch_bwd_nt<T, 2>(q, g, o, d, s);
        case 3: return launch_bwd_nt<T, 3>(q, g, o, d, s);
        case 4: return launch_bwd_nt<T, 4>(q, g, o, d, s);
        case 5: return launch_bwd_nt<T, 5>(q, g, o, d, s);
        case 6: return launch_bwd_nt<T, 6>(q, g, o, d, s);
        case 7: return launch_bwd_nt<T, 7>(q, g, o, d, s);
        case 8: return launch_bwd_nt<T, 8>(q, g, o, d, s);
        default: MIVIT_FAIL("attention_bwd: S=%d > 128 is not supported by the LDS-resident kernel", S);
    }
}

}  // namespace

int attention_max_seq(int dtype, int Dh) {
    int best = 0;
    for (int S = 1; S <= 128; ++S) {
        const AttnDims d = dtype == MIVIT_F32 ? make_dims<float>(1, S, 1, Dh, true) : make_dims<bf16>(1, S, 1, Dh, true);
        if ((size_t)d.per_wave * dtype_size(dtype) <= LDS_LIMIT || attention_fast_supported(dtype, S, Dh) ||
            attention_fast_supported_f16(dtype, S, Dh)) best = S;
    }
    return best;
}

int launch_attention_fwd(int dtype, const void *qkv, int B, int S, int H, int Dh, void *ctx, hipStream_t s) {
    MIVIT_CHECK(B > 0 && S > 0 && H > 0 && Dh > 0, "attention_fwd: empty problem");
    if (attention_fast_supported(dtype, S, Dh)) return launch_attention_fwd_fast(qkv, B, S, H, Dh, ctx, s);
    if (attention_fast_supported_f16(dtype, S, Dh)) return launch_attention_fwd_fast_f16(qkv, B, S, H, Dh, ctx, s);     // (elem.h: the same unit compiled for IEEE half)
    return dtype == MIVIT_F32 ? attn_fwd_t<float>(qkv, B, S, H, Dh, ctx, s)
         : dtype == MIVIT_BF16 ? attn_fwd_t<bf16>(qkv, B, S, H, Dh, ctx, s) : attn_fwd_t<f16>(qkv, B, S, H, Dh, ctx, s);
}
int launch_attention_bwd(int dtype, const void *qkv, const void *dctx, int B, int S, int H, int Dh, void *dqkv,
                         hipStream_t s) {
    MIVIT_CHECK(B > 0 && S > 0 && H > 0 && Dh > 0, "attention_bwd: empty problem");
    if (attention_fast_supported(dtype, S, Dh)) return launch_attention_bwd_fast(qkv, dctx, B, S, H, Dh, dqkv, s);
    if (attention_fast_supported_f16(dtype, S, Dh)) return launch_attention_bwd_fast_f16(qkv, dctx, B, S, H, Dh, dqkv, s);
    return dtype == MIVIT_F32 ? attn_bwd_t<float>(qkv, dctx, B, S, H, Dh, dqkv, s)
         : dtype == MIVIT_BF16 ? attn_bwd_t<bf16>(qkv, dctx, B, S, H, Dh, dqkv, s) : attn_bwd_t<f16>(qkv, dctx, B, S, H, Dh, dqkv, s);
}

extern "C" int mivit_attention_max_seq(int dtype, int Dh) { return attention_max_seq(dtype, Dh); }
extern "C" int mivit_attention_fwd(int dtype, const void *qkv, int B, int S, int H, int Dh, void *ctx, void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16 || dtype == MIVIT_F16, "bad dtype %d", dtype);
    MIVIT_CHECK(qkv && ctx, "attention_fwd: null pointer");
    return launch_attention_fwd(dtype, qkv, B, S, H, Dh, ctx, static_cast<hipStream_t>(stream));
}
extern "C" int mivit_attention_bwd(int dtype, const void *qkv, const void *dctx, int B, int S, int H, int Dh,
                                   void *dqkv, void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16 || dtype == MIVIT_F16, "bad dtype %d", dtype);
    MIVIT_CHECK(qkv && dctx && dqkv, "attention_bwd: null pointer");
    return launch_attention_bwd(dtype, qkv, dctx, B, S, H, Dh, dqkv, static_cast<hipStream_t>(stream));
}
