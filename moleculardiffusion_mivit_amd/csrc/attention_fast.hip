// bf16 fast path of the attention core for short sequences (S <= 64, head dim 16/32: every shipped MiViT config).
//
// One 64-lane wavefront per (batch, head).  What makes it cheap:
//  * Q/K/V/dO MFMA fragments whose contraction index is the head dimension are loaded STRAIGHT from HBM/L2 into
//    registers (16 contiguous bytes per lane): no LDS staging, no zero fill, and the same registers serve as A or
//    B operand (both want [row or col = lane&15][k = 8*(lane>>4)..+7]).
//  * Products that contract over tokens take one operand from the accumulator registers of the score tile
//    (C layout: lane = column, registers = rows) with a consistent permutation of the k index, and the other
//    through ds_read_b64_tr_b16 from a NATURAL [token][d] LDS image: no transposed copies, no P / dS round trip.
//      forward : S^T = K Q^T (lane = query row) -> softmax in-lane + 2 shuffles -> O = P V   (V image in LDS)
//      backward: S = Q K^T, dP = dO V^T (lane = key)  -> dV += P^T dO, dK += dS^T Q           (dO, Q images)
//                S^T, dP^T            (lane = query)  -> dQ  = dS K                            (K image)
//    (scores are computed in both orientations in backward: 4 extra MFMAs per tile pair buy zero LDS traffic
//    for the probabilities).
// Everything else (fp32 mode, long sequences, wide heads) uses the general kernels in attention.hip.
#include "common.h"
#include <stdlib.h>
#include "stream_prims.h"
#include "elem.h"          // element type of this translation unit (bf16, or f16 under -DMIVIT_ELEM_F16): after every other include

namespace {


struct U128 {
    uint32_t w[4];
};

__device__ __forceinline__ bf16x8 as_frag(const U128 &u) { return __builtin_bit_cast(bf16x8, u); }

// fragment straight from global memory: element (row0 + lane&15, d0 + 8*(lane>>4) .. +7); zero outside [S) x [Dh)
__device__ __forceinline__ bf16x8 gfrag(const bf16 *base, int64_t ld, int row0, int S, int d0, int Dh, int lane) {
    const int r = row0 + (lane & 15), d = d0 + 8 * (lane >> 4);
    U128 u = {{0u, 0u, 0u, 0u}};
    if (r < S && d + 8 <= Dh) {
        const uint4 v = *reinterpret_cast<const uint4 *>(base + (int64_t)r * ld + d);
        u.w[0] = v.x; u.w[1] = v.y; u.w[2] = v.z; u.w[3] = v.w;
    }
    return as_frag(u);
}

// B (or A) fragment whose k index runs over tokens, read transposed from a natural [token][d] LDS image:
// k slots 0..3 = tokens ta + 0..3, k slots 4..7 = tokens tb + 0..3 (ta, tb already include the lane group's 4*g),
// column = col0 + (lane & 15).
__device__ __forceinline__ bf16x8 tr_frag(const bf16 *img, int ld, int ta, int tb, int col0, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3;
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(img + (ta + q) * ld + col0 + 4 * p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(img + (tb + q) * ld + col0 + 4 * p));
    struct { s16x4 a, b; } pr = {lo, hi};
    return __builtin_bit_cast(bf16x8, pr);
}

// 8 fp32 accumulator values (two tiles' 4 rows each) -> one bf16 operand fragment
__device__ __forceinline__ bf16x8 pack_frag(const f32x4 a, const f32x4 b) {
    bf16x8 f;
    f[0] = (__bf16)a[0]; f[1] = (__bf16)a[1]; f[2] = (__bf16)a[2]; f[3] = (__bf16)a[3];
    f[4] = (__bf16)b[0]; f[5] = (__bf16)b[1]; f[6] = (__bf16)b[2]; f[7] = (__bf16)b[3];
    return f;
}


// reductions over the 16 lanes of a row / across the 4 rows: DPP and permlane-swap forms (common.h), no LDS round trips
__device__ __forceinline__ float g16_max(float v) { return row16_max(v); }
__device__ __forceinline__ float g16_sum(float v) { return row16_sum(v); }
__device__ __forceinline__ float x4_max(float v) { return rows4_max(v); }   // across the 4 lane groups (same lane & 15)
__device__ __forceinline__ float x4_sum(float v) { return rows4_sum(v); }

__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// stage a natural [S][Dh] head slice into an LDS image of NTP*16 rows (rows >= S zero), 16-byte chunks
// SWZ (Dh = 32 only: 64-byte rows, no padding): the two 32-byte halves of rows 4..7 (mod 8) are exchanged.  A transposing
// read (two 32-lane groups, 64 banks) touches 8 consecutive rows x 32 bytes; rows r and r + 4 of an unpadded 64-byte-row
// image start on the same banks, so with the exchange they use opposite halves of their 16-bank window: conflict-free,
// where the padded 80-byte pitch was 2-way on every such read (SQ_LDS_BANK_CONFLICT 0.24 of the LDS-active cycles).
template <bool SWZ = false>
__device__ __forceinline__ void stage_nat(bf16 *img, int ld, const bf16 *src, int64_t src_ld, int S, int rows, int Dh,
                                          int lane) {
    const int cpr = Dh >> 3, n = rows * cpr;
    // four chunks in flight per lane: one load -> wait -> store per iteration (what this run-time loop compiled to) costs a
    // memory round trip per 64 chunks -- 30 of them in a row for three 80-row x 64-wide images
    for (int u0 = lane; u0 < n; u0 += 4 * 64) {
        uint4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = u0 + 64 * i, r = u / cpr, d = (u % cpr) * 8;
            v[i] = make_uint4(0u, 0u, 0u, 0u);
            if (u < n && r < S) v[i] = *reinterpret_cast<const uint4 *>(src + (int64_t)r * src_ld + d);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = u0 + 64 * i, r = u / cpr, d = (u % cpr) * 8;
            if (u < n) *reinterpret_cast<uint4 *>(img + r * ld + (SWZ ? (d ^ (((r >> 2) & 1) << 4)) : d)) = v[i];
        }
    }
}

struct FastDims {
    int B, S, H, Dh, E;
    int ld;          // LDS image row length (elements)
    int img;         // elements per image (NTP * 16 * ld)
};

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
template <int NT, int ND>
__global__ __launch_bounds__(256) void attn_fwd_fast(const bf16 *qkv, bf16 *ctx, const FastDims d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int KD = (ND + 1) / 2;          // 32-wide k steps over the head dimension
    constexpr int NP = (NT + 1) / 2;          // token tile pairs (32-wide k steps over tokens)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * (blockDim.x >> 6) + wave;
    if (pair >= d.B * d.H) return;
    const int b = pair / d.H, h = pair % d.H;
    const int g = lane >> 4, cq = lane & 15;
    bf16 *Vimg = reinterpret_cast<bf16 *>(smem_raw) + (size_t)wave * d.img;
    const int64_t ld3 = 3 * (int64_t)d.E;
    const bf16 *q = qkv + (int64_t)b * d.S * ld3 + h * d.Dh, *k = q + d.E, *v = q + 2 * d.E;
    stage_nat(Vimg, d.ld, v, ld3, d.S, NP * 32, d.Dh, lane);
    bf16x8 kf[NT][KD];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) kf[j][kd] = gfrag(k, ld3, j * 16, d.S, kd * 32, d.Dh, lane);
    lds_sync();
    const float scale = 1.0f / sqrtf((float)d.Dh);

    for (int it = 0; it < NT; ++it) {
        bf16x8 qf[KD];
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) qf[kd] = gfrag(q, ld3, it * 16, d.S, kd * 32, d.Dh, lane);
        // S^T tiles: rows = keys (registers), column = query row it*16 + cq (lane)
        f32x4 st[NT];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            st[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) st[j] = mma(kf[j][kd], qf[kd], st[j]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = j * 16 + 4 * g + r < d.S;
                st[j][r] = ok ? st[j][r] * scale : -INFINITY;
                m = fmaxf(m, st[j][r]);
            }
        }
        m = x4_max(m);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = (j * 16 + 4 * g + r < d.S) ? __expf(st[j][r] - m) : 0.f;
                st[j][r] = e;
                sum += e;
            }
        const float inv = 1.f / x4_sum(sum);
#pragma unroll
        for (int j = 0; j < NT; ++j) st[j] *= inv;
        // O[row][d] = sum_key P[row][key] V[key][d]: A = P from registers (lane = row), B = V image transposed-read
#pragma unroll
        for (int jd = 0; jd < ND; ++jd) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NP; ++ks) {
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                const bf16x8 a = pack_frag(st[2 * ks], (2 * ks + 1 < NT) ? st[2 * ks + 1] : zero);
                const bf16x8 bb = tr_frag(Vimg, d.ld, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, jd * 16, lane);
                o = mma(a, bb, o);
            }
            // C layout: column = d (jd*16 + cq), rows = query it*16 + 4g + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = it * 16 + 4 * g + r;
                if (row < d.S) ctx[((int64_t)b * d.S + row) * d.E + h * d.Dh + jd * 16 + cq] = from_f32<bf16>(o[r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------
template <int NT, int ND>
__global__ __launch_bounds__(256) void attn_bwd_fast(const bf16 *qkv, const bf16 *dctx, bf16 *dqkv, const FastDims d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int KD = (ND + 1) / 2;
    constexpr int NP = (NT + 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * (blockDim.x >> 6) + wave;
    if (pair >= d.B * d.H) return;
    const int b = pair / d.H, h = pair % d.H;
    const int g = lane >> 4, cq = lane & 15;
    bf16 *Qimg = reinterpret_cast<bf16 *>(smem_raw) + (size_t)wave * 3 * d.img;
    bf16 *Kimg = Qimg + d.img, *Oimg = Kimg + d.img;
    const int64_t ld3 = 3 * (int64_t)d.E;
    const bf16 *q = qkv + (int64_t)b * d.S * ld3 + h * d.Dh, *k = q + d.E, *v = q + 2 * d.E;
    const bf16 *dO = dctx + (int64_t)b * d.S * d.E + h * d.Dh;
    stage_nat(Qimg, d.ld, q, ld3, d.S, NP * 32, d.Dh, lane);
    stage_nat(Kimg, d.ld, k, ld3, d.S, NP * 32, d.Dh, lane);
    stage_nat(Oimg, d.ld, dO, d.E, d.S, NP * 32, d.Dh, lane);
    bf16x8 kf[NT][KD], vf[NT][KD], qf[NT][KD], of[NT][KD];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) {
            kf[j][kd] = gfrag(k, ld3, j * 16, d.S, kd * 32, d.Dh, lane);
            vf[j][kd] = gfrag(v, ld3, j * 16, d.S, kd * 32, d.Dh, lane);
            qf[j][kd] = gfrag(q, ld3, j * 16, d.S, kd * 32, d.Dh, lane);
            of[j][kd] = gfrag(dO, d.E, j * 16, d.S, kd * 32, d.Dh, lane);
        }
    lds_sync();
    const float scale = 1.0f / sqrtf((float)d.Dh);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    bf16 *dst = dqkv + (int64_t)b * d.S * ld3 + h * d.Dh;

    f32x4 dv[NT][ND], dk[NT][ND];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int jd = 0; jd < ND; ++jd) { dv[j][jd] = zero; dk[j][jd] = zero; }

#pragma unroll
    for (int ip = 0; ip < NP; ++ip) {
        f32x4 pU[2][NT], sU[2][NT];     // P and dS of the two row tiles of this pair, lane = key, registers = rows
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int it = 2 * ip + t;
            if (it >= NT) {
#pragma unroll
                for (int j = 0; j < NT; ++j) { pU[t][j] = zero; sU[t][j] = zero; }
                continue;
            }
            // ---- lane = key orientation: S = Q K^T, dP = dO V^T ----
            f32x4 sc[NT], dp[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                sc[j] = zero; dp[j] = zero;
#pragma unroll
                for (int kd = 0; kd < KD; ++kd) {
                    sc[j] = mma(qf[it][kd], kf[j][kd], sc[j]);
                    dp[j] = mma(of[it][kd], vf[j][kd], dp[j]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const bool ok = j * 16 + cq < d.S;
                    sc[j][r] = ok ? sc[j][r] * scale : -INFINITY;
                    m = fmaxf(m, sc[j][r]);
                }
                m = g16_max(m);
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float e = (j * 16 + cq < d.S) ? __expf(sc[j][r] - m) : 0.f;
                    sc[j][r] = e;
                    sum += e;
                }
                const float inv = 1.f / g16_sum(sum);
                float delta = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    sc[j][r] *= inv;
                    delta += sc[j][r] * dp[j][r];
                }
                delta = g16_sum(delta);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    pU[t][j][r] = sc[j][r];
                    sU[t][j][r] = scale * sc[j][r] * (dp[j][r] - delta);
                }
            }
            // ---- lane = query orientation: S^T = K Q^T, dP^T = V dO^T -> dQ for this row tile ----
            f32x4 st[NT], dt[NT];
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                st[j] = zero; dt[j] = zero;
#pragma unroll
                for (int kd = 0; kd < KD; ++kd) {
                    st[j] = mma(kf[j][kd], qf[it][kd], st[j]);
                    dt[j] = mma(vf[j][kd], of[it][kd], dt[j]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = j * 16 + 4 * g + r < d.S;
                    st[j][r] = ok ? st[j][r] * scale : -INFINITY;
                    m = fmaxf(m, st[j][r]);
                }
            }
            m = x4_max(m);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = (j * 16 + 4 * g + r < d.S) ? __expf(st[j][r] - m) : 0.f;
                    st[j][r] = e;
                    sum += e;
                }
            const float inv = 1.f / x4_sum(sum);
            float delta = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[j][r] *= inv;
                    delta += st[j][r] * dt[j][r];
                }
            delta = x4_sum(delta);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[j][r] = scale * st[j][r] * (dt[j][r] - delta);    // dS^T
#pragma unroll
            for (int jd = 0; jd < ND; ++jd) {
                f32x4 dq = zero;
#pragma unroll
                for (int ks = 0; ks < NP; ++ks) {
                    const bf16x8 a = pack_frag(st[2 * ks], (2 * ks + 1 < NT) ? st[2 * ks + 1] : zero);
                    const bf16x8 bb = tr_frag(Kimg, d.ld, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, jd * 16, lane);
                    dq = mma(a, bb, dq);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = it * 16 + 4 * g + r;
                    if (row < d.S) dst[(int64_t)row * ld3 + jd * 16 + cq] = from_f32<bf16>(dq[r]);
                }
            }
        }
        // ---- dV += P^T dO, dK += dS^T Q over the 32 query rows of this pair ----
#pragma unroll
        for (int jd = 0; jd < ND; ++jd) {
            const bf16x8 bo = tr_frag(Oimg, d.ld, 32 * ip + 4 * g, 32 * ip + 16 + 4 * g, jd * 16, lane);
            const bf16x8 bq = tr_frag(Qimg, d.ld, 32 * ip + 4 * g, 32 * ip + 16 + 4 * g, jd * 16, lane);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                dv[j][jd] = mma(pack_frag(pU[0][j], pU[1][j]), bo, dv[j][jd]);
                dk[j][jd] = mma(pack_frag(sU[0][j], sU[1][j]), bq, dk[j][jd]);
            }
        }
    }
    // C layout of dV / dK tiles: column = d (jd*16 + cq), rows = key j*16 + 4g + r
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int jd = 0; jd < ND; ++jd)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = j * 16 + 4 * g + r;
                if (key < d.S) {
                    bf16 *o = dst + (int64_t)key * ld3 + jd * 16 + cq;
                    o[d.E] = from_f32<bf16>(dk[j][jd][r]);
                    o[2 * d.E] = from_f32<bf16>(dv[j][jd][r]);
                }
            }
}

// ---------------------------------------------------------------------------------------------------------------
// backward, second version: the token contractions (dV += P^T dO, dK += dS^T Q, dQ += dS K) use the 16-deep MFMA
// (v_mfma_f32_16x16x16_bf16): the accumulator layout of a 16 x 16 score tile (rows 4g + r) IS the k layout of that
// instruction's operand (k = 4g + i), so one row tile is consumed at a time -- no pairing of row tiles into 32-deep
// operands, no P / dS tiles parked across two iterations, no padding work for odd tile counts -- and the Q / dO fragments
// are fetched per row tile instead of being held for the whole sequence.  Fewer live registers = more waves per SIMD;
// this kernel is latency-bound (1.8 TB/s of traffic), not bandwidth-bound.
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short k16_t;
__device__ __forceinline__ k16_t pack4(const f32x4 a) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 f;
    f[0] = (__bf16)a[0]; f[1] = (__bf16)a[1]; f[2] = (__bf16)a[2]; f[3] = (__bf16)a[3];
    return __builtin_bit_cast(k16_t, f);
}
// operand with k = tokens t0 + 4g + 0..3 (t0 already includes 4g), lane index = column col0 + (lane & 15)
template <bool SWZ = false>
__device__ __forceinline__ k16_t tr4(const bf16 *img, int ld, int trow, int col0, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3;
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const int row = trow + q, col = col0 + 4 * p;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(img + row * ld + (SWZ ? (col ^ (((row >> 2) & 1) << 4)) : col)));
}
__device__ __forceinline__ f32x4 mma16(k16_t a, k16_t b, f32x4 c) {
    return ELEM_MFMA_16x16x16(a, b, c);
}

template <int NT, int ND>
__global__ __launch_bounds__(256, (NT * ND <= 6 ? 3 : 1)) void attn_bwd_fast2(const bf16 *qkv, const bf16 *dctx, bf16 *dqkv, const FastDims d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int KD = (ND + 1) / 2;
    constexpr bool SWZ = ND == 2;          // 64-byte rows: unpadded images with the half-row exchange (launcher sets d.ld = 32)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * (blockDim.x >> 6) + wave;
    if (pair >= d.B * d.H) return;
    const int b = pair / d.H, h = pair % d.H;
    const int g = lane >> 4, cq = lane & 15;
    bf16 *Qimg = reinterpret_cast<bf16 *>(smem_raw) + (size_t)wave * (3 * d.img + NT * 256);
    bf16 *Kimg = Qimg + d.img, *Oimg = Kimg + d.img, *Simg = Oimg + d.img;     // Simg: dS of the current row tile, [key][16 queries]
    const int64_t ld3 = 3 * (int64_t)d.E;
    const bf16 *q = qkv + (int64_t)b * d.S * ld3 + h * d.Dh, *k = q + d.E, *v = q + 2 * d.E;
    const bf16 *dO = dctx + (int64_t)b * d.S * d.E + h * d.Dh;
    // Every global load of the prologue is issued before the first wait.  (The staging helper's loop has a run-time trip
    // count: left as three calls it compiled to load -> s_waitcnt vmcnt(0) -> ds_write per 64 chunks, i.e. NINE serialised
    // memory round trips at the start of every wave, and the per-row-tile q / dO fragment loads added three more.)
    constexpr int CPR = ND * 2, NCH = NT * 16 * CPR, NIT = (NCH + 63) / 64;       // 16-byte chunks per image, per lane
    constexpr bool BATCH = NIT <= 4 && NT <= 4;      // (longer sequences hold more accumulators: no room for the staging registers)
    bf16x8 kf[NT][KD], vf[NT][KD];
    if (BATCH) {
        uint4 sq[NIT], sk[NIT], so[NIT];
        auto ld_nat = [&](const bf16 *src, int64_t sld, uint4 (&v)[NIT]) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int u = lane + 64 * i, r = u / CPR, dd = (u % CPR) * 8;
                v[i] = make_uint4(0u, 0u, 0u, 0u);
                if (u < NCH && r < d.S && dd + 8 <= d.Dh) v[i] = *reinterpret_cast<const uint4 *>(src + (int64_t)r * sld + dd);
            }
        };
        auto st_nat = [&](bf16 *img, const uint4 (&v)[NIT]) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int u = lane + 64 * i, r = u / CPR, dd = (u % CPR) * 8;
                if (u < NCH) *reinterpret_cast<uint4 *>(img + r * d.ld + (SWZ ? (dd ^ (((r >> 2) & 1) << 4)) : dd)) = v[i];
            }
        };
        ld_nat(q, ld3, sq); ld_nat(k, ld3, sk); ld_nat(dO, d.E, so);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) {
                vf[j][kd] = gfrag(v, ld3, j * 16, d.S, kd * 32, d.Dh, lane);
                if (!SWZ) kf[j][kd] = gfrag(k, ld3, j * 16, d.S, kd * 32, d.Dh, lane);
            }
        st_nat(Qimg, sq); st_nat(Kimg, sk); st_nat(Oimg, so);
    } else {
        stage_nat<SWZ>(Qimg, d.ld, q, ld3, d.S, NT * 16, d.Dh, lane);
        stage_nat<SWZ>(Kimg, d.ld, k, ld3, d.S, NT * 16, d.Dh, lane);
        stage_nat<SWZ>(Oimg, d.ld, dO, d.E, d.S, NT * 16, d.Dh, lane);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) {
                vf[j][kd] = gfrag(v, ld3, j * 16, d.S, kd * 32, d.Dh, lane);
                if (!SWZ) kf[j][kd] = gfrag(k, ld3, j * 16, d.S, kd * 32, d.Dh, lane);
            }
    }
    lds_sync();
    // SWZ (Dh = 32): row-operand fragments come out of the images just staged (one conflict-free ds_read_b128 each: with the
    // half-row exchange the four row quads of a b128 lane group use the four different 16-byte slots), not from global memory
    auto lfrag = [&](const bf16 *img, int row0) {
        const int r = row0 + cq, col = 8 * g;
        return *reinterpret_cast<const bf16x8 *>(img + r * d.ld + (col ^ (((r >> 2) & 1) << 4)));
    };
    if (SWZ) {
#pragma unroll
        for (int j = 0; j < NT; ++j) kf[j][0] = lfrag(Kimg, j * 16);
    }
    const float scale = 1.0f / sqrtf((float)d.Dh);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    bf16 *dst = dqkv + (int64_t)b * d.S * ld3 + h * d.Dh;

    f32x4 dv[NT][ND], dk[NT][ND];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int jd = 0; jd < ND; ++jd) { dv[j][jd] = zero; dk[j][jd] = zero; }

#pragma unroll 1
    for (int it = 0; it < NT; ++it) {
        bf16x8 qf[KD], of[KD];
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) {
            if (SWZ) { qf[kd] = lfrag(Qimg, it * 16); of[kd] = lfrag(Oimg, it * 16); }
            else {
                qf[kd] = gfrag(q, ld3, it * 16, d.S, kd * 32, d.Dh, lane);
                of[kd] = gfrag(dO, d.E, it * 16, d.S, kd * 32, d.Dh, lane);
            }
        }
        f32x4 dS[NT];        // dS of this row tile: lane = key j*16 + cq, registers = queries it*16 + 4g + r
        {   // ---- lane = key orientation: S = Q K^T, dP = dO V^T  ->  P, dS of this row tile  ->  dV, dK ----
            f32x4 sc[NT], dp[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                sc[j] = zero; dp[j] = zero;
#pragma unroll
                for (int kd = 0; kd < KD; ++kd) {
                    sc[j] = mma(qf[kd], kf[j][kd], sc[j]);
                    dp[j] = mma(of[kd], vf[j][kd], dp[j]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const bool ok = j * 16 + cq < d.S;
                    sc[j][r] = ok ? sc[j][r] * scale : -INFINITY;
                    m = fmaxf(m, sc[j][r]);
                }
                m = g16_max(m);
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float e = (j * 16 + cq < d.S) ? __expf(sc[j][r] - m) : 0.f;
                    sc[j][r] = e;
                    sum += e;
                }
                const float inv = 1.f / g16_sum(sum);
                float delta = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    sc[j][r] *= inv;
                    delta += sc[j][r] * dp[j][r];
                }
                delta = g16_sum(delta);
#pragma unroll
                for (int j = 0; j < NT; ++j) dp[j][r] = scale * sc[j][r] * (dp[j][r] - delta);     // dS
            }
            // C layout of P / dS tile j: column = key j*16 + cq, rows = query it*16 + 4g + r  ==  A operand [key][query k]
            // of the 16-deep MFMA is its TRANSPOSE: A[row = key][k = query] -- which is what lane cq = key, k = 4g + r holds
#pragma unroll
            for (int jd = 0; jd < ND; ++jd) {
                const k16_t bo = tr4<SWZ>(Oimg, d.ld, it * 16 + 4 * g, jd * 16, lane);
                const k16_t bq = tr4<SWZ>(Qimg, d.ld, it * 16 + 4 * g, jd * 16, lane);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    dv[j][jd] = mma16(pack4(sc[j]), bo, dv[j][jd]);
                    dk[j][jd] = mma16(pack4(dp[j]), bq, dk[j][jd]);
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) dS[j] = dp[j];
        }
        {   // ---- dQ of this row tile = dS K: the contraction runs over keys, which sit on the LANES of the dS tiles above.
            //      One trip through a wave-private LDS image turns them: each lane stores its 4 consecutive queries of key
            //      j*16 + cq (8 bytes) into a [key][query] image, a transposing read returns lane = query, k slots = keys --
            //      instead of recomputing scores, dP and the softmax in the other orientation (12 MFMAs + a second softmax pass).
            lds_sync();            // the previous row tile's dS reads are done
#pragma unroll
            for (int j = 0; j < NT; ++j) *reinterpret_cast<k16_t *>(Simg + (j * 16 + cq) * 16 + 4 * g) = pack4(dS[j]);
            lds_sync();
#pragma unroll
            for (int jd = 0; jd < ND; ++jd) {
                f32x4 dq = zero;
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    dq = mma16(tr4(Simg, 16, j * 16 + 4 * g, 0, lane), tr4<SWZ>(Kimg, d.ld, j * 16 + 4 * g, jd * 16, lane), dq);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = it * 16 + 4 * g + r;
                    if (row < d.S) dst[(int64_t)row * ld3 + jd * 16 + cq] = from_f32<bf16>(dq[r]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int jd = 0; jd < ND; ++jd)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = j * 16 + 4 * g + r;
                if (key < d.S) {
                    bf16 *o = dst + (int64_t)key * ld3 + jd * 16 + cq;
                    o[d.E] = from_f32<bf16>(dk[j][jd][r]);
                    o[2 * d.E] = from_f32<bf16>(dv[j][jd][r]);
                }
            }
}

FastDims make_fast(int B, int S, int H, int Dh) {
    FastDims d;
    d.B = B; d.S = S; d.H = H; d.Dh = Dh; d.E = H * Dh;
    const int NT = (S + 15) / 16, NP = (NT + 1) / 2;
    d.ld = Dh + 8;
    d.img = NP * 32 * d.ld;
    return d;
}

// waves per workgroup: 4, fewer when the per-wave LDS images would not fit 160 KB
static int waves_per_block(size_t per_wave_bytes) {
    const int w = (int)((size_t)160 * 1024 / per_wave_bytes);
    return w < 1 ? 1 : (w > 4 ? 4 : w);
}

template <int NT, int ND>
int fwd_launch(const bf16 *qkv, bf16 *ctx, const FastDims &d, hipStream_t s) {
    const size_t per_wave = (size_t)d.img * sizeof(bf16);
    const int wpb = waves_per_block(per_wave);
    auto kern = attn_fwd_fast<NT, ND>;
    if (per_wave * wpb > 64 * 1024)
        MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(per_wave * wpb)));
    ProfScope prof(s);
    hipLaunchKernelGGL(kern, dim3(ceil_div(d.B * d.H, wpb)), dim3(64 * wpb), per_wave * wpb, s, qkv, ctx, d);
    MIVIT_LAUNCH_CHECK();
    return 0;
}
template <int NT, int ND>
int bwd_launch(const bf16 *qkv, const bf16 *dctx, bf16 *dqkv, const FastDims &d, hipStream_t s) {
    const size_t per_wave = (size_t)3 * d.img * sizeof(bf16);
    const int wpb = waves_per_block(per_wave);
    static const int version = getenv("MIVIT_ATTN_BWD") ? atoi(getenv("MIVIT_ATTN_BWD")) : 2;     // 1 = first (pair-tiled) version
    auto kern = version == 2 ? attn_bwd_fast2<NT, ND> : attn_bwd_fast<NT, ND>;
    if (version == 2) {        // the 16-deep version needs NT * 16 image rows, not the pair-padded NP * 32: more waves fit a CU
        FastDims d2 = d;
        if (ND == 2) d2.ld = 32;             // unpadded rows + half-row exchange (conflict-free transposing reads)
        d2.img = NT * 16 * d2.ld;
        const size_t pw = ((size_t)3 * d2.img + NT * 256) * sizeof(bf16);
        const int w2 = waves_per_block(pw);       // (15 waves per CU in 3-wave workgroups at <= 128 registers: 25 spilled registers, 1.44 -> 2.25 ms per step)
        if (pw * w2 > 64 * 1024)
            MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(pw * w2)));
        ProfScope prof(s);
        hipLaunchKernelGGL(kern, dim3(ceil_div(d.B * d.H, w2)), dim3(64 * w2), pw * w2, s, qkv, dctx, dqkv, d2);
        MIVIT_LAUNCH_CHECK();
        return 0;
    }
    if (per_wave * wpb > 64 * 1024)
        MIVIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(per_wave * wpb)));
    ProfScope prof(s);
    hipLaunchKernelGGL(kern, dim3(ceil_div(d.B * d.H, wpb)), dim3(64 * wpb), per_wave * wpb, s, qkv, dctx, dqkv, d);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

}  // namespace

bool attention_fast_supported(int dtype, int S, int Dh) {
    // one wavefront holds the whole (batch, head) problem in registers: the backward's register budget sets the limit
    if (dtype != MIVIT_ELEM_DTYPE || S < 1) return false;
    // (up to 5 token tiles at Dh = 64 / 7 at 32 / 8 at 16 compile without scratch; the longer ones spill a little in the
    //  backward but keep every sequence the reference's 128-entry positional table allows on this path)
    const int NT = (S + 15) / 16;
    return (Dh == 16 || Dh == 32 || Dh == 64) && NT <= 8;
}

#define FAST_DISPATCH(FN, ...)                                                 \
    switch (NT * 10 + ND) {                                                    \
        case 11: return FN<1, 1>(__VA_ARGS__);                                 \
        case 21: return FN<2, 1>(__VA_ARGS__);                                 \
        case 31: return FN<3, 1>(__VA_ARGS__);                                 \
        case 41: return FN<4, 1>(__VA_ARGS__);                                 \
        case 12: return FN<1, 2>(__VA_ARGS__);                                 \
        case 22: return FN<2, 2>(__VA_ARGS__);                                 \
        case 32: return FN<3, 2>(__VA_ARGS__);                                 \
        case 42: return FN<4, 2>(__VA_ARGS__);                                 \
        case 14: return FN<1, 4>(__VA_ARGS__);                                 \
        case 24: return FN<2, 4>(__VA_ARGS__);                                 \
        case 34: return FN<3, 4>(__VA_ARGS__);                                 \
        case 44: return FN<4, 4>(__VA_ARGS__);                                 \
        case 51: return FN<5, 1>(__VA_ARGS__);                                 \
        case 52: return FN<5, 2>(__VA_ARGS__);                                 \
        case 54: return FN<5, 4>(__VA_ARGS__);                                 \
        case 61: return FN<6, 1>(__VA_ARGS__);                                 \
        case 62: return FN<6, 2>(__VA_ARGS__);                                 \
        case 71: return FN<7, 1>(__VA_ARGS__);                                 \
        case 72: return FN<7, 2>(__VA_ARGS__);                                 \
        case 81: return FN<8, 1>(__VA_ARGS__);                                 \
        case 82: return FN<8, 2>(__VA_ARGS__);                                 \
        case 64: return FN<6, 4>(__VA_ARGS__);                                 \
        case 74: return FN<7, 4>(__VA_ARGS__);                                 \
        case 84: return FN<8, 4>(__VA_ARGS__);                                 \
        default: MIVIT_FAIL("attention fast path: unsupported tile shape");    \
    }

int launch_attention_fwd_fast(const void *qkv, int B, int S, int H, int Dh, void *ctx, hipStream_t s) {
    const FastDims d = make_fast(B, S, H, Dh);
    const int NT = (S + 15) / 16, ND = Dh / 16;
    FAST_DISPATCH(fwd_launch, static_cast<const bf16 *>(qkv), static_cast<bf16 *>(ctx), d, s)
}

int launch_attention_bwd_fast(const void *qkv, const void *dctx, int B, int S, int H, int Dh, void *dqkv, hipStream_t s) {
    const FastDims d = make_fast(B, S, H, Dh);
    const int NT = (S + 15) / 16, ND = Dh / 16;
    FAST_DISPATCH(bwd_launch, static_cast<const bf16 *>(qkv), static_cast<const bf16 *>(dctx), static_cast<bf16 *>(dqkv), d, s)
}
