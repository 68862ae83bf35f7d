// Tiled MFMA GEMM for every dense contraction on the MiViT path (gfx950).
//
//   C[m,n] = epilogue( sum_k A(m,k) * B(n,k) )
//
// A and B are staged global -> registers -> LDS as [row][k] images in the compute type T (bf16 or f32), whatever
// their layout / element type in HBM, so the MFMA fragment reads are identical for the three uses:
//   forward  y  = x W^T      : A = x  (k contiguous),   B = W [N,K]      (k contiguous)
//   dgrad    dx = dy W       : A = dy (k contiguous),   B = W [N,K]      (k strided: register-block transpose)
//   wgrad    dW = dy^T x     : A = dy (k = rows, strided), B = x (k = rows, strided), split over rows into slabs
// fp32 master weights are converted to bf16 on the way into LDS (no shadow copies to keep coherent).
#include "common.h"

namespace {

enum { KC = 0 /* src[r * ld + k] */, KSTR = 1 /* src[k * ld + r] */ };

struct GemmArgs {
    const void *A; int64_t lda;
    const void *B; int64_t ldb;
    int M, N, K;
    int k_chunk;                 // reduction range handled by one blockIdx.z
    const float *bias;           // [N] or null
    int act;                     // activation applied to acc + bias
    const void *dact; int64_t ldd; int dact_kind;   // multiply by act'(dact[m,n])
    const void *resid; int64_t ldr;                 // + resid[m,n]  (T)
    void *C; int64_t ldc; int c_is_f32;
    void *C2;                    // optional pre-activation output (T, ldc)
    int map_rows, map_stride, map_off;
    int64_t slab_stride;         // floats between the per-z slabs (split reduction)
    int accumulate;              // C += result (fp32 C only)
};

template <typename T>
struct Cfg {
    // bf16: BK = 64, LDS rows of 160 B (ds_read_b128 fragment reads conflict-free: slot = (10*row + chunk) mod 16
    // separates the even/odd halves of every 16-lane read group);  fp32: BK = 32, rows of 144 B.
    static constexpr int BK = (sizeof(T) == 2) ? 64 : 32;
    static constexpr int LDK = (sizeof(T) == 2) ? 80 : 36;
};

template <typename TS> struct VecOf;
template <> struct VecOf<float> { typedef float4 type; };
template <int K> struct VecOf<h16<K>> { typedef uint4 type; };

// raw staged vector of source type TS -> floats
template <typename TS>
__device__ __forceinline__ void unpack(const typename VecOf<TS>::type &v, float *out) {
    if constexpr (sizeof(TS) == 4) {
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else if constexpr (sizeof(TS) == 2 && __is_same(TS, bf16)) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            out[2 * i] = __uint_as_float(w[i] << 16);
            out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    } else {
        const f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] = (float)h[i];
    }
}
// element-wise tail / unaligned path: zero fill beyond nvalid (static register indexing only: no scratch)
__device__ __forceinline__ float4 load_vec_guarded(const float *p, int nvalid) {
    float4 v;
    v.x = nvalid > 0 ? p[0] : 0.f;
    v.y = nvalid > 1 ? p[1] : 0.f;
    v.z = nvalid > 2 ? p[2] : 0.f;
    v.w = nvalid > 3 ? p[3] : 0.f;
    return v;
}
template <int K>
__device__ __forceinline__ uint4 load_vec_guarded(const h16<K> *p, int nvalid) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t lo = nvalid > 2 * i ? (uint32_t)p[2 * i].v : 0u;
        const uint32_t hi = nvalid > 2 * i + 1 ? (uint32_t)p[2 * i + 1].v : 0u;
        w[i] = lo | (hi << 16);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <typename T>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    return (uint32_t)from_f32<T>(lo).v | ((uint32_t)from_f32<T>(hi).v << 16);
}
template <typename T, int N>
__device__ __forceinline__ void store_lds(T *dst, const float *v) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        if constexpr (N == 8) *reinterpret_cast<float4 *>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else if constexpr (N == 4) {
        *reinterpret_cast<uint2 *>(dst) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
    } else {
        *reinterpret_cast<uint4 *>(dst) =
            make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7]));
    }
}

// Two-phase staging: load() issues the global loads of one k-tile into registers (kept raw, so they can stay in
// flight under the MFMAs of the previous tile), store() converts and writes the [row][k] LDS image.
// k-contiguous source: image[r][k] <- src[(r0 + r) * ld + k0 + k]
template <typename T, typename TS, int R>
struct StageKC {
    static constexpr int V = 16 / sizeof(TS);
    static constexpr int TPR = Cfg<T>::BK / V;
    static constexpr int RPP = 256 / TPR;
    static constexpr int NP = R / RPP;
    static_assert(R % RPP == 0, "tile rows must be a multiple of the rows staged per pass");
    typename VecOf<TS>::type regs[NP];
    __device__ __forceinline__ void load(const TS *src, int64_t ld, int r0, int rmax, int k0, int kend, bool vec_ok,
                                         int tid) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int gr = r0 + p * RPP + tid / TPR, gk = k0 + (tid % TPR) * V;
            if (gr < rmax && gk + V <= kend && vec_ok) {
                regs[p] = *reinterpret_cast<const typename VecOf<TS>::type *>(src + (int64_t)gr * ld + gk);
            } else {
                const int nv = gr < rmax ? max(0, min(V, kend - gk)) : 0;
                regs[p] = load_vec_guarded(src + (int64_t)(gr < rmax ? gr : 0) * ld + gk, nv);
            }
        }
    }
    __device__ __forceinline__ void store(T *img, int tid) const {
        constexpr int LDK = Cfg<T>::LDK;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float v[V];
            unpack<TS>(regs[p], v);
            store_lds<T, V>(img + (p * RPP + tid / TPR) * LDK + (tid % TPR) * V, v);
        }
    }
};

// k-strided source: image[r][k] <- src[(k0 + k) * ld + r0 + r]; a unit = KB k-rows x V consecutive r, transposed
// in registers, written as V LDS rows of KB consecutive k (one 16-byte store each).
template <typename T, typename TS, int R>
struct StageKS {
    static constexpr int V = 16 / sizeof(TS);
    static constexpr int KB = 16 / sizeof(T);
    static constexpr int UR = R / V;
    static constexpr int UK = Cfg<T>::BK / KB;
    static constexpr int UNITS = UR * UK;
    static_assert(UNITS <= 256, "one staging unit per thread");
    typename VecOf<TS>::type regs[KB];
    __device__ __forceinline__ void load(const TS *src, int64_t ld, int r0, int rmax, int k0, int kend, bool vec_ok,
                                         int tid) {
        if (tid >= UNITS) return;
        const int gr = r0 + (tid % UR) * V, kg = (tid / UR) * KB;
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const int gk = k0 + kg + i;
            if (gk < kend && gr + V <= rmax && vec_ok) {
                regs[i] = *reinterpret_cast<const typename VecOf<TS>::type *>(src + (int64_t)gk * ld + gr);
            } else {
                const int nv = gk < kend ? max(0, min(V, rmax - gr)) : 0;
                regs[i] = load_vec_guarded(src + (int64_t)(gk < kend ? gk : 0) * ld + gr, nv);
            }
        }
    }
    __device__ __forceinline__ void store(T *img, int tid) const {
        constexpr int LDK = Cfg<T>::LDK;
        if (tid >= UNITS) return;
        const int rv = (tid % UR) * V, kg = (tid / UR) * KB;
        float v[KB][V];
#pragma unroll
        for (int i = 0; i < KB; ++i) unpack<TS>(regs[i], v[i]);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float t[KB];
#pragma unroll
            for (int i = 0; i < KB; ++i) t[i] = v[i][j];
            store_lds<T, KB>(img + (rv + j) * LDK + kg, t);
        }
    }
};

template <typename T, typename TS, int L, int R> struct StageSel;
template <typename T, typename TS, int R> struct StageSel<T, TS, KC, R> { typedef StageKC<T, TS, R> type; };
template <typename T, typename TS, int R> struct StageSel<T, TS, KSTR, R> { typedef StageKS<T, TS, R> type; };

template <typename T, typename TA, int LA, typename TB, int LB, int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
    constexpr int BK = Cfg<T>::BK, LDK = Cfg<T>::LDK;
    constexpr int TM = BM / 32, TN = BN / 32;   // 16x16 tiles per wave (waves arranged 2 x 2)
    __shared__ __attribute__((aligned(16))) T smem[(BM + BN) * LDK];
    T *As = smem, *Bs = smem + BM * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kb = blockIdx.z * g.k_chunk;
    const int ke = min(g.K, kb + g.k_chunk);

    const TA *A = static_cast<const TA *>(g.A);
    const TB *B = static_cast<const TB *>(g.B);
    constexpr int VA = 16 / sizeof(TA), VB = 16 / sizeof(TB);
    const bool va_ok = (g.lda % VA == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool vb_ok = (g.ldb % VB == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    typename StageSel<T, TA, LA, BM>::type sa;
    typename StageSel<T, TB, LB, BN>::type sb;
    if (kb < ke) {
        sa.load(A, g.lda, m0, g.M, kb, ke, va_ok, tid);
        sb.load(B, g.ldb, n0, g.N, kb, ke, vb_ok, tid);
        sa.store(As, tid);
        sb.store(Bs, tid);
    }
    __syncthreads();
    for (int k0 = kb; k0 < ke; k0 += BK) {
        const bool more = k0 + BK < ke;
        if (more) {   // next tile's global loads fly under this tile's MFMAs
            sa.load(A, g.lda, m0, g.M, k0 + BK, ke, va_ok, tid);
            sb.load(B, g.ldb, n0, g.N, k0 + BK, ke, vb_ok, tid);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += Mma<T>::KS) {
            typename Mma<T>::Frag a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Mma<T>::load(As, LDK, 1, wm * (BM / 2) + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Mma<T>::load(Bs, LDK, 1, wn * (BN / 2) + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::mma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
        if (more) {
            sa.store(As, tid);
            sb.store(Bs, tid);
            __syncthreads();
        }
    }

    const int colq = lane & 15, rowq = (lane >> 4) * 4;
    if (g.c_is_f32) {
        // ---- fp32 output (fp32 mode, split-reduction slabs, model output): straight from the accumulator layout
        float *Cf = static_cast<float *>(g.C) + (int64_t)blockIdx.z * g.slab_stride;
        T *C2 = static_cast<T *>(g.C2);
        const T *resid = static_cast<const T *>(g.resid);
        const T *dact = static_cast<const T *>(g.dact);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (BN / 2) + j * 16 + colq;
                if (col >= g.N) continue;
                const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * (BM / 2) + i * 16 + rowq + r;
                    if (row >= g.M) continue;
                    float v = acc[i][j][r] + bv;
                    int64_t orow = row;
                    if (g.map_rows > 0) orow = (int64_t)(row / g.map_rows) * g.map_stride + row % g.map_rows + g.map_off;
                    if (C2) C2[orow * g.ldc + col] = from_f32<T>(v);
                    v = act_fwd(g.act, v);
                    if (dact) v *= act_bwd(g.dact_kind, to_f32(dact[(int64_t)row * g.ldd + col]));
                    if (resid) v += to_f32(resid[(int64_t)row * g.ldr + col]);
                    float *p = Cf + orow * g.ldc + col;
                    *p = g.accumulate ? (*p + v) : v;
                }
            }
        }
        return;
    }

    // ---- bf16 output: stage acc + bias through LDS as fp32 (half the tile rows at a time), then every thread
    //      finishes 8 consecutive columns: 16-byte loads of residual / saved activation, one rounding, 16-byte store.
    if constexpr (sizeof(T) == 2) {
        constexpr int HR = BM / 2;          // rows per pass = the rows owned by the waves with wm == h
        constexpr int LDC = BN + 4;         // fp32 row stride of the staging tile (16-byte aligned rows)
        static_assert((size_t)HR * LDC * 4 <= (size_t)(BM + BN) * LDK * sizeof(T), "staging tile must fit in the operand LDS");
        float *Cs = reinterpret_cast<float *>(smem);
        T *Ct = static_cast<T *>(g.C);
        T *C2 = static_cast<T *>(g.C2);
        const T *resid = static_cast<const T *>(g.resid);
        const T *dact = static_cast<const T *>(g.dact);
        const bool vec_c = (g.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(Ct) & 15) == 0) &&
                           (!C2 || (reinterpret_cast<uintptr_t>(C2) & 15) == 0) &&
                           (!resid || (g.ldr % 8 == 0 && (reinterpret_cast<uintptr_t>(resid) & 15) == 0)) &&
                           (!dact || (g.ldd % 8 == 0 && (reinterpret_cast<uintptr_t>(dact) & 15) == 0));
        for (int h = 0; h < 2; ++h) {
            if (h) __syncthreads();
            if (wm == h) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int lc = wn * (BN / 2) + j * 16 + colq;
                        const float bv = (g.bias && n0 + lc < g.N) ? g.bias[n0 + lc] : 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) Cs[(i * 16 + rowq + r) * LDC + lc] = acc[i][j][r] + bv;
                    }
            }
            __syncthreads();
            for (int c = tid; c < HR * (BN / 8); c += 256) {
                const int lr = c / (BN / 8), lc = (c % (BN / 8)) * 8;
                const int row = m0 + h * HR + lr, col = n0 + lc;
                if (row >= g.M || col >= g.N) continue;
                int64_t orow = row;
                if (g.map_rows > 0) orow = (int64_t)(row / g.map_rows) * g.map_stride + row % g.map_rows + g.map_off;
                float v[8];
                *reinterpret_cast<float4 *>(v) = *reinterpret_cast<const float4 *>(Cs + lr * LDC + lc);
                *reinterpret_cast<float4 *>(v + 4) = *reinterpret_cast<const float4 *>(Cs + lr * LDC + lc + 4);
                if (col + 8 <= g.N && vec_c) {
                    if (C2) store_lds<T, 8>(C2 + orow * g.ldc + col, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = act_fwd(g.act, v[e]);
                    if (dact) {
                        float d[8];
                        unpack<T>(*reinterpret_cast<const uint4 *>(dact + (int64_t)row * g.ldd + col), d);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= act_bwd(g.dact_kind, d[e]);
                    }
                    if (resid) {
                        float d[8];
                        unpack<T>(*reinterpret_cast<const uint4 *>(resid + (int64_t)row * g.ldr + col), d);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += d[e];
                    }
                    store_lds<T, 8>(Ct + orow * g.ldc + col, v);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (col + e < g.N) {
                            float x = v[e];
                            if (C2) C2[orow * g.ldc + col + e] = from_f32<T>(x);
                            x = act_fwd(g.act, x);
                            if (dact) x *= act_bwd(g.dact_kind, to_f32(dact[(int64_t)row * g.ldd + col + e]));
                            if (resid) x += to_f32(resid[(int64_t)row * g.ldr + col + e]);
                            Ct[orow * g.ldc + col + e] = from_f32<T>(x);
                        }
                    }
                }
            }
        }
    }
}

template <typename T, typename TA, int LA, typename TB, int LB>
int launch_gemm_t(const GemmArgs &g, int splits, hipStream_t s) {
    // 64x64 tiles when the 128x128 grid would leave most of the 256 CUs idle
    const long big = (long)ceil_div(g.M, 128) * ceil_div(g.N, 128) * splits;
    ProfScope prof(s);
    if (sizeof(TA) == 4 && sizeof(T) == 2 && LA == KC && g.K >= 1024 && g.N >= 128 && big >= 192) {
        // streaming fp32 rows (frame embedding): a 64-row tile halves the prefetch registers -> 3+ blocks per CU
        // keep enough bytes in flight to cover HBM latency
        dim3 grid(ceil_div(g.N, 128), ceil_div(g.M, 64), splits);
        hipLaunchKernelGGL((gemm_kernel<T, TA, LA, TB, LB, 64, 128>), grid, dim3(256), 0, s, g);
    } else if (sizeof(TA) == 4 && sizeof(T) == 2 && sizeof(TB) == 4 && LA == KC) {
        // (fp32 activations AND fp32 weights both staged through registers: the 128x128 tile would spill)
        dim3 grid(ceil_div(g.N, 64), ceil_div(g.M, 64), splits);
        hipLaunchKernelGGL((gemm_kernel<T, TA, LA, TB, LB, 64, 64>), grid, dim3(256), 0, s, g);
    } else if (big >= 192 || (g.M > 64 && g.N > 64 && big >= 64)) {
        dim3 grid(ceil_div(g.N, 128), ceil_div(g.M, 128), splits);
        hipLaunchKernelGGL((gemm_kernel<T, TA, LA, TB, LB, 128, 128>), grid, dim3(256), 0, s, g);
    } else {
        dim3 grid(ceil_div(g.N, 64), ceil_div(g.M, 64), splits);
        hipLaunchKernelGGL((gemm_kernel<T, TA, LA, TB, LB, 64, 64>), grid, dim3(256), 0, s, g);
    }
    MIVIT_LAUNCH_CHECK();
    return 0;
}

// column sums of dy[M,N] over a row chunk: part[chunk][n].  Thread = V consecutive columns (16-byte loads) x one of
// 8 row lanes; 8-way LDS reduction per block.
template <typename TS>
__global__ __launch_bounds__(256) void colsum_kernel(const TS *src, int64_t ld, int M, int N, int chunk, float *part) {
    constexpr int V = 16 / sizeof(TS);
    __shared__ float red[8][32 * V];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = (blockIdx.x * 32 + tx) * V;
    const int rb = blockIdx.y * chunk, re = min(M, rb + chunk);
    const bool vec_ok = (ld % V == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (c0 + V <= N);
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f;
    if (c0 < N) {
        for (int r = rb + ty; r < re; r += 8) {
            float v[V];
            if (vec_ok) unpack<TS>(*reinterpret_cast<const typename VecOf<TS>::type *>(src + (int64_t)r * ld + c0), v);
            else
#pragma unroll
                for (int i = 0; i < V; ++i) v[i] = (c0 + i < N) ? to_f32(src[(int64_t)r * ld + c0 + i]) : 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] += v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) red[ty][tx * V + i] = acc[i];
    __syncthreads();
    for (int c = threadIdx.x; c < 32 * V; c += 256) {
        const int col = blockIdx.x * 32 * V + c;
        if (col < N) {
            float t = 0.f;
#pragma unroll
            for (int y = 0; y < 8; ++y) t += red[y][c];
            part[(int64_t)blockIdx.y * N + col] = t;
        }
    }
}

int wgrad_splits(int M, int N, int K) {
    // total blocks ~ 1.5 per CU: enough to stream from every CU, few enough that the fp32 slabs
    // (splits * N * K * 4 bytes written + re-read) stay a fraction of the operand traffic
    const long tiles = (long)ceil_div(N, 128) * ceil_div(K, 128);
    long want = (384 + tiles - 1) / tiles;
    long maxs = (M + 255) / 256;                          // at least 256 reduction rows per split
    long s = want < maxs ? want : maxs;
    if (s < 1) s = 1;
    if (s > 512) s = 512;
    return (int)s;
}
int colsum_chunks(int M) {
    if (M <= 256) return 1;                          // tiny: one chunk, written straight into the result (no reduce launch)
    int c = ceil_div(M, M < 8192 ? 64 : 512);       // short problems: more, shorter row chunks (the row loop is a latency chain)
    return c < 1 ? 1 : (c > 256 ? 256 : c);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
template <typename H>
static int linear_fwd_h(const LinearFwdArgs &a, const GemmArgs &g, hipStream_t s) {
    if (a.w_is_bf16) {      // weights already in the 16-bit compute type (per-step shadow copy)
        if (a.x_is_f32) return launch_gemm_t<H, float, KC, H, KC>(g, 1, s);
        return launch_gemm_t<H, H, KC, H, KC>(g, 1, s);
    }
    if (a.x_is_f32) return launch_gemm_t<H, float, KC, float, KC>(g, 1, s);
    return launch_gemm_t<H, H, KC, float, KC>(g, 1, s);
}

int launch_linear_fwd(const LinearFwdArgs &a, hipStream_t s) {
    MIVIT_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "linear_fwd: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    GemmArgs g = {};
    g.A = a.x; g.lda = a.ldx; g.B = a.W; g.ldb = a.K;
    g.M = a.M; g.N = a.N; g.K = a.K; g.k_chunk = (a.K + 63) / 64 * 64;
    g.bias = a.bias; g.act = a.act;
    g.resid = a.resid; g.ldr = a.ldr;
    g.C = a.y; g.ldc = a.ldy; g.c_is_f32 = (a.dtype == MIVIT_F32) || a.y_is_f32;
    g.C2 = a.y_preact;
    g.map_rows = a.map_rows; g.map_stride = a.map_stride; g.map_off = a.map_off;
    if (a.dtype == MIVIT_F32) return launch_gemm_t<float, float, KC, float, KC>(g, 1, s);
    return a.dtype == MIVIT_BF16 ? linear_fwd_h<bf16>(a, g, s) : linear_fwd_h<f16>(a, g, s);
}

int launch_linear_dgrad(const LinearDgradArgs &a, hipStream_t s) {
    MIVIT_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "linear_dgrad: empty problem");
    GemmArgs g = {};
    g.A = a.dy; g.lda = a.lddy; g.B = a.W; g.ldb = a.K;      // B(n = k_in, k = n_out) = W[n_out * K + k_in]
    g.M = a.M; g.N = a.K; g.K = a.N; g.k_chunk = (a.N + 63) / 64 * 64;
    g.dact = a.act != MIVIT_ACT_NONE ? a.saved : nullptr; g.ldd = a.lds; g.dact_kind = a.act;
    g.resid = a.dres; g.ldr = a.lddr;
    g.C = a.dx; g.ldc = a.lddx; g.c_is_f32 = (a.dtype == MIVIT_F32) || a.dx_is_f32;
    if (a.dtype == MIVIT_F32) return launch_gemm_t<float, float, KC, float, KSTR>(g, 1, s);
    MIVIT_CHECK(!a.dy_is_f32, "linear_dgrad: fp32 dy in 16-bit mode is not instantiated (convert first)");
    if (a.dtype == MIVIT_BF16) {
        if (a.w_is_bf16) return launch_gemm_t<bf16, bf16, KC, bf16, KSTR>(g, 1, s);
        return launch_gemm_t<bf16, bf16, KC, float, KSTR>(g, 1, s);
    }
    if (a.w_is_bf16) return launch_gemm_t<f16, f16, KC, f16, KSTR>(g, 1, s);
    return launch_gemm_t<f16, f16, KC, float, KSTR>(g, 1, s);
}

size_t linear_wgrad_ws_bytes(int M, int N, int K) {
    const int sp = wgrad_splits(M, N, K);
    size_t b = (sp > 1 ? (size_t)sp * N * K * sizeof(float) : 0);
    b += (size_t)colsum_chunks(M) * N * sizeof(float);
    return align_up(b, 256);
}

int launch_linear_wgrad(const LinearWgradArgs &a, hipStream_t s) {
    MIVIT_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "linear_wgrad: empty problem");
    MIVIT_CHECK(a.ws_bytes >= linear_wgrad_ws_bytes(a.M, a.N, a.K), "linear_wgrad: workspace too small");
    const int sp = wgrad_splits(a.M, a.N, a.K);
    float *slabs = static_cast<float *>(a.ws);
    float *bias_part = slabs + (sp > 1 ? (size_t)sp * a.N * a.K : 0);
    if (a.dW) {
        GemmArgs g = {};
        g.A = a.dy; g.lda = a.lddy; g.B = a.x; g.ldb = a.ldx;
        g.M = a.N; g.N = a.K; g.K = a.M;
        g.k_chunk = (ceil_div(a.M, sp) + 63) / 64 * 64;
        const int splits = ceil_div(a.M, g.k_chunk);
        g.c_is_f32 = 1; g.ldc = a.K;
        if (splits > 1) { g.C = slabs; g.slab_stride = (int64_t)a.N * a.K; }
        else { g.C = a.dW; g.accumulate = a.accumulate; }
        int rc;
        if (a.dtype == MIVIT_F32) rc = launch_gemm_t<float, float, KSTR, float, KSTR>(g, splits, s);
        else if (a.dy_is_f32) { MIVIT_FAIL("linear_wgrad: fp32 dy in 16-bit mode is not instantiated (convert first)"); }
        else if (a.dtype == MIVIT_BF16 && a.x_is_f32) rc = launch_gemm_t<bf16, bf16, KSTR, float, KSTR>(g, splits, s);
        else if (a.dtype == MIVIT_BF16) rc = launch_gemm_t<bf16, bf16, KSTR, bf16, KSTR>(g, splits, s);
        else if (a.x_is_f32) rc = launch_gemm_t<f16, f16, KSTR, float, KSTR>(g, splits, s);
        else rc = launch_gemm_t<f16, f16, KSTR, f16, KSTR>(g, splits, s);
        if (rc) return rc;
        if (splits > 1) {
            rc = launch_slab_reduce(slabs, splits, (int64_t)a.N * a.K, a.dW, a.accumulate, s);
            if (rc) return rc;
        }
    }
    if (a.db) {
        const int chunks = colsum_chunks(a.M);
        const int chunk = ceil_div(a.M, chunks);
        const bool direct = chunks == 1 && !a.accumulate;       // the single partial row IS the result
        if (direct) bias_part = a.db;
        const int cols_per_block = (a.dtype == MIVIT_F32 || a.dy_is_f32) ? 32 * 4 : 32 * 8;
        dim3 grid(ceil_div(a.N, cols_per_block), chunks);
        if (a.dtype == MIVIT_F32 || a.dy_is_f32)
            hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, static_cast<const float *>(a.dy), a.lddy,
                               a.M, a.N, chunk, bias_part);
        else if (a.dtype == MIVIT_BF16)
            hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, s, static_cast<const bf16 *>(a.dy), a.lddy,
                               a.M, a.N, chunk, bias_part);
        else
            hipLaunchKernelGGL(colsum_kernel<f16>, grid, dim3(256), 0, s, static_cast<const f16 *>(a.dy), a.lddy,
                               a.M, a.N, chunk, bias_part);
        MIVIT_LAUNCH_CHECK();
        if (!direct) {
            int rc = launch_slab_reduce(bias_part, chunks, a.N, a.db, a.accumulate, s);
            if (rc) return rc;
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C-ABI (operator level)
// ------------------------------------------------------------------------------------------------
extern "C" int mivit_linear_fwd(int dtype, const void *x, int x_is_f32, int64_t ldx, const float *W,
                                const float *bias, int M, int N, int K, int act, const void *resid, int64_t ldr,
                                void *y, int64_t ldy, void *y_preact, void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16 || dtype == MIVIT_F16, "bad dtype %d", dtype);
    MIVIT_CHECK(x && W && y, "linear_fwd: null pointer");
    LinearFwdArgs a = {};
    a.dtype = dtype; a.x = x; a.x_is_f32 = x_is_f32 || dtype == MIVIT_F32; a.ldx = ldx; a.W = W; a.w_is_bf16 = 0; a.bias = bias;
    a.M = M; a.N = N; a.K = K; a.act = act; a.resid = resid; a.ldr = ldr; a.y = y; a.ldy = ldy; a.y_preact = y_preact;
    return launch_linear_fwd(a, static_cast<hipStream_t>(stream));
}

extern "C" int mivit_linear_dgrad(int dtype, const void *dy, int64_t lddy, const float *W, int M, int N, int K,
                                  int act, const void *saved, int64_t lds, const void *dres, int64_t lddr,
                                  void *dx, int64_t lddx, void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16 || dtype == MIVIT_F16, "bad dtype %d", dtype);
    MIVIT_CHECK(dy && W && dx, "linear_dgrad: null pointer");
    MIVIT_CHECK(act == MIVIT_ACT_NONE || saved, "linear_dgrad: activation backward needs `saved`");
    LinearDgradArgs a = {};
    a.dtype = dtype; a.dy = dy; a.dy_is_f32 = dtype == MIVIT_F32; a.lddy = lddy; a.W = W; a.M = M; a.N = N; a.K = K;
    a.act = act; a.saved = saved; a.lds = lds; a.dres = dres; a.lddr = lddr; a.dx = dx; a.lddx = lddx;
    return launch_linear_dgrad(a, static_cast<hipStream_t>(stream));
}

extern "C" size_t mivit_linear_wgrad_workspace_bytes(int M, int N, int K) { return linear_wgrad_ws_bytes(M, N, K); }

extern "C" int mivit_linear_wgrad(int dtype, const void *dy, int64_t lddy, const void *x, int x_is_f32, int64_t ldx,
                                  int M, int N, int K, float *dW, float *db, int accumulate, void *workspace,
                                  size_t workspace_bytes, void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16 || dtype == MIVIT_F16, "bad dtype %d", dtype);
    MIVIT_CHECK(dy && x && workspace, "linear_wgrad: null pointer");
    LinearWgradArgs a = {};
    a.dtype = dtype; a.dy = dy; a.dy_is_f32 = dtype == MIVIT_F32; a.lddy = lddy;
    a.x = x; a.x_is_f32 = x_is_f32 || dtype == MIVIT_F32; a.ldx = ldx;
    a.M = M; a.N = N; a.K = K; a.dW = dW; a.db = db; a.accumulate = accumulate; a.ws = workspace; a.ws_bytes = workspace_bytes;
    return launch_linear_wgrad(a, static_cast<hipStream_t>(stream));
}
