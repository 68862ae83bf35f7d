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
    static constexpr int BK = 32;
    static constexpr int LDK = (sizeof(T) == 2) ? 48 : 36;   // padded LDS row, elements (96 B / 144 B)
};

template <typename TS>
__device__ __forceinline__ void load_vec(const TS *p, float *out);
template <>
__device__ __forceinline__ void load_vec<float>(const float *p, float *out) {
    const float4 v = *reinterpret_cast<const float4 *>(p);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <>
__device__ __forceinline__ void load_vec<bf16>(const bf16 *p, float *out) {
    const uint4 v = *reinterpret_cast<const uint4 *>(p);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        out[2 * i] = __uint_as_float(w[i] << 16);
        out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}

template <typename T, int N>
__device__ __forceinline__ void store_lds(T *dst, const float *v);
template <>
__device__ __forceinline__ void store_lds<float, 4>(float *dst, const float *v) {
    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void store_lds<float, 8>(float *dst, const float *v) {
    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4 *>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    return (uint32_t)from_f32<bf16>(lo).v | ((uint32_t)from_f32<bf16>(hi).v << 16);
}
template <>
__device__ __forceinline__ void store_lds<bf16, 4>(bf16 *dst, const float *v) {
    *reinterpret_cast<uint2 *>(dst) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
}
template <>
__device__ __forceinline__ void store_lds<bf16, 8>(bf16 *dst, const float *v) {
    *reinterpret_cast<uint4 *>(dst) =
        make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
}

// k-contiguous source: image[r][k] <- src[(r0 + r) * ld + k0 + k]
template <typename T, typename TS, int R>
__device__ __forceinline__ void stage_kc(T *img, const TS *src, int64_t ld, int r0, int rmax, int k0, int kend,
                                         bool vec_ok, int tid) {
    constexpr int V = 16 / sizeof(TS);
    constexpr int TPR = Cfg<T>::BK / V;
    constexpr int RPP = 256 / TPR;
    constexpr int LDK = Cfg<T>::LDK;
    static_assert(R % RPP == 0, "tile rows must be a multiple of the rows staged per pass");
#pragma unroll
    for (int p = 0; p < R / RPP; ++p) {
        const int r = p * RPP + tid / TPR;
        const int kv = (tid % TPR) * V;
        const int gr = r0 + r, gk = k0 + kv;
        float v[V];
        if (gr < rmax && gk + V <= kend && vec_ok) {
            load_vec<TS>(src + (int64_t)gr * ld + gk, v);
        } else {
#pragma unroll
            for (int i = 0; i < V; ++i)
                v[i] = (gr < rmax && gk + i < kend) ? to_f32(src[(int64_t)gr * ld + gk + i]) : 0.f;
        }
        store_lds<T, V>(img + r * LDK + kv, v);
    }
}

// k-strided source: image[r][k] <- src[(k0 + k) * ld + r0 + r]; each unit loads KB k-rows of V consecutive r,
// transposes in registers and writes V rows of KB consecutive k (one 16-byte LDS store each).
template <typename T, typename TS, int R>
__device__ __forceinline__ void stage_ks(T *img, const TS *src, int64_t ld, int r0, int rmax, int k0, int kend,
                                         bool vec_ok, int tid) {
    constexpr int V = 16 / sizeof(TS);
    constexpr int KB = 16 / sizeof(T);
    constexpr int UR = R / V;
    constexpr int UK = Cfg<T>::BK / KB;
    constexpr int UNITS = UR * UK;
    constexpr int LDK = Cfg<T>::LDK;
    for (int u = tid; u < UNITS; u += 256) {
        const int rv = (u % UR) * V, kg = (u / UR) * KB;
        const int gr = r0 + rv;
        float v[KB][V];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const int gk = k0 + kg + i;
            if (gk < kend && gr + V <= rmax && vec_ok) {
                load_vec<TS>(src + (int64_t)gk * ld + gr, v[i]);
            } else {
#pragma unroll
                for (int j = 0; j < V; ++j)
                    v[i][j] = (gk < kend && gr + j < rmax) ? to_f32(src[(int64_t)gk * ld + gr + j]) : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float t[KB];
#pragma unroll
            for (int i = 0; i < KB; ++i) t[i] = v[i][j];
            store_lds<T, KB>(img + (rv + j) * LDK + kg, t);
        }
    }
}

template <typename T, typename TA, int LA, typename TB, int LB, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g) {
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

    for (int k0 = kb; k0 < ke; k0 += BK) {
        if (LA == KC) stage_kc<T, TA, BM>(As, A, g.lda, m0, g.M, k0, ke, va_ok, tid);
        else          stage_ks<T, TA, BM>(As, A, g.lda, m0, g.M, k0, ke, va_ok, tid);
        if (LB == KC) stage_kc<T, TB, BN>(Bs, B, g.ldb, n0, g.N, k0, ke, vb_ok, tid);
        else          stage_ks<T, TB, BN>(Bs, B, g.ldb, n0, g.N, k0, ke, vb_ok, tid);
        __syncthreads();
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
    }

    // ---- epilogue: lane holds column (lane & 15), rows (lane >> 4) * 4 + j of each 16x16 tile ----
    float *Cf = static_cast<float *>(g.C) + (int64_t)blockIdx.z * g.slab_stride;
    T *Ct = static_cast<T *>(g.C);
    T *C2 = static_cast<T *>(g.C2);
    const T *resid = static_cast<const T *>(g.resid);
    const T *dact = static_cast<const T *>(g.dact);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 16 + (lane & 15);
            if (col >= g.N) continue;
            const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * (BM / 2) + i * 16 + (lane >> 4) * 4 + r;
                if (row >= g.M) continue;
                float v = acc[i][j][r] + bv;
                int64_t orow = row;
                if (g.map_rows > 0) orow = (int64_t)(row / g.map_rows) * g.map_stride + row % g.map_rows + g.map_off;
                if (C2) C2[orow * g.ldc + col] = from_f32<T>(v);
                v = act_fwd(g.act, v);
                if (dact) v *= act_bwd(g.dact_kind, to_f32(dact[(int64_t)row * g.ldd + col]));
                if (resid) v += to_f32(resid[(int64_t)row * g.ldr + col]);
                if (g.c_is_f32) {
                    float *p = Cf + orow * g.ldc + col;
                    *p = g.accumulate ? (*p + v) : v;
                } else {
                    Ct[orow * g.ldc + col] = from_f32<T>(v);
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
    if (big >= 192 || (g.M > 64 && g.N > 64 && big >= 64)) {
        dim3 grid(ceil_div(g.N, 128), ceil_div(g.M, 128), splits);
        hipLaunchKernelGGL((gemm_kernel<T, TA, LA, TB, LB, 128, 128>), grid, dim3(256), 0, s, g);
    } else {
        dim3 grid(ceil_div(g.N, 64), ceil_div(g.M, 64), splits);
        hipLaunchKernelGGL((gemm_kernel<T, TA, LA, TB, LB, 64, 64>), grid, dim3(256), 0, s, g);
    }
    MIVIT_LAUNCH_CHECK();
    return 0;
}

// column sums of dy[M,N] over a row chunk: part[z][n]
template <typename TS>
__global__ __launch_bounds__(256) void colsum_kernel(const TS *src, int64_t ld, int M, int N, int chunk, float *part) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int rb = blockIdx.y * chunk, re = min(M, rb + chunk);
    float acc = 0.f;
    if (c < N)
        for (int r = rb + rl; r < re; r += 4) acc += to_f32(src[(int64_t)r * ld + c]);
    red[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && c < N)
        part[(int64_t)blockIdx.y * N + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

int wgrad_splits(int M, int N, int K) {
    const long tiles = (long)ceil_div(N, 128) * ceil_div(K, 128);
    long want = (1024 + tiles - 1) / tiles;               // ~4 blocks per CU in flight
    long maxs = (M + 255) / 256;                          // at least 256 reduction rows per split
    long s = want < maxs ? want : maxs;
    if (s < 1) s = 1;
    if (s > 512) s = 512;
    return (int)s;
}
int colsum_chunks(int M) {
    int c = ceil_div(M, 512);
    return c < 1 ? 1 : (c > 256 ? 256 : c);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
int launch_linear_fwd(const LinearFwdArgs &a, hipStream_t s) {
    MIVIT_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "linear_fwd: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    GemmArgs g = {};
    g.A = a.x; g.lda = a.ldx; g.B = a.W; g.ldb = a.K;
    g.M = a.M; g.N = a.N; g.K = a.K; g.k_chunk = (a.K + 31) / 32 * 32;
    g.bias = a.bias; g.act = a.act;
    g.resid = a.resid; g.ldr = a.ldr;
    g.C = a.y; g.ldc = a.ldy; g.c_is_f32 = (a.dtype == MIVIT_F32) || a.y_is_f32;
    g.C2 = a.y_preact;
    g.map_rows = a.map_rows; g.map_stride = a.map_stride; g.map_off = a.map_off;
    if (a.dtype == MIVIT_F32) return launch_gemm_t<float, float, KC, float, KC>(g, 1, s);
    if (a.x_is_f32) return launch_gemm_t<bf16, float, KC, float, KC>(g, 1, s);
    return launch_gemm_t<bf16, bf16, KC, float, KC>(g, 1, s);
}

int launch_linear_dgrad(const LinearDgradArgs &a, hipStream_t s) {
    MIVIT_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "linear_dgrad: empty problem");
    GemmArgs g = {};
    g.A = a.dy; g.lda = a.lddy; g.B = a.W; g.ldb = a.K;      // B(n = k_in, k = n_out) = W[n_out * K + k_in]
    g.M = a.M; g.N = a.K; g.K = a.N; g.k_chunk = (a.N + 31) / 32 * 32;
    g.dact = a.act != MIVIT_ACT_NONE ? a.saved : nullptr; g.ldd = a.lds; g.dact_kind = a.act;
    g.resid = a.dres; g.ldr = a.lddr;
    g.C = a.dx; g.ldc = a.lddx; g.c_is_f32 = (a.dtype == MIVIT_F32) || a.dx_is_f32;
    if (a.dtype == MIVIT_F32) return launch_gemm_t<float, float, KC, float, KSTR>(g, 1, s);
    MIVIT_CHECK(!a.dy_is_f32, "linear_dgrad: fp32 dy in bf16 mode is not instantiated (convert first)");
    return launch_gemm_t<bf16, bf16, KC, float, KSTR>(g, 1, s);
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
        g.k_chunk = (ceil_div(a.M, sp) + 31) / 32 * 32;
        const int splits = ceil_div(a.M, g.k_chunk);
        g.c_is_f32 = 1; g.ldc = a.K;
        if (splits > 1) { g.C = slabs; g.slab_stride = (int64_t)a.N * a.K; }
        else { g.C = a.dW; g.accumulate = a.accumulate; }
        int rc;
        if (a.dtype == MIVIT_F32) rc = launch_gemm_t<float, float, KSTR, float, KSTR>(g, splits, s);
        else if (a.dy_is_f32) { MIVIT_FAIL("linear_wgrad: fp32 dy in bf16 mode is not instantiated (convert first)"); }
        else if (a.x_is_f32) rc = launch_gemm_t<bf16, bf16, KSTR, float, KSTR>(g, splits, s);
        else rc = launch_gemm_t<bf16, bf16, KSTR, bf16, KSTR>(g, splits, s);
        if (rc) return rc;
        if (splits > 1) {
            rc = launch_slab_reduce(slabs, splits, (int64_t)a.N * a.K, a.dW, a.accumulate, s);
            if (rc) return rc;
        }
    }
    if (a.db) {
        const int chunks = colsum_chunks(a.M);
        const int chunk = ceil_div(a.M, chunks);
        dim3 grid(ceil_div(a.N, 64), chunks);
        if (a.dtype == MIVIT_F32 || a.dy_is_f32)
            hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, static_cast<const float *>(a.dy), a.lddy,
                               a.M, a.N, chunk, bias_part);
        else
            hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, s, static_cast<const bf16 *>(a.dy), a.lddy,
                               a.M, a.N, chunk, bias_part);
        MIVIT_LAUNCH_CHECK();
        int rc = launch_slab_reduce(bias_part, chunks, a.N, a.db, a.accumulate, s);
        if (rc) return rc;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C-ABI (operator level)
// ------------------------------------------------------------------------------------------------
extern "C" int mivit_linear_fwd(int dtype, const void *x, int x_is_f32, int64_t ldx, const float *W,
                                const float *bias, int M, int N, int K, int act, const void *resid, int64_t ldr,
                                void *y, int64_t ldy, void *y_preact, void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16, "bad dtype %d", dtype);
    MIVIT_CHECK(x && W && y, "linear_fwd: null pointer");
    LinearFwdArgs a = {};
    a.dtype = dtype; a.x = x; a.x_is_f32 = x_is_f32 || dtype == MIVIT_F32; a.ldx = ldx; a.W = W; a.bias = bias;
    a.M = M; a.N = N; a.K = K; a.act = act; a.resid = resid; a.ldr = ldr; a.y = y; a.ldy = ldy; a.y_preact = y_preact;
    return launch_linear_fwd(a, static_cast<hipStream_t>(stream));
}

extern "C" int mivit_linear_dgrad(int dtype, const void *dy, int64_t lddy, const float *W, int M, int N, int K,
                                  int act, const void *saved, int64_t lds, const void *dres, int64_t lddr,
                                  void *dx, int64_t lddx, void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16, "bad dtype %d", dtype);
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
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16, "bad dtype %d", dtype);
    MIVIT_CHECK(dy && x && workspace, "linear_wgrad: null pointer");
    LinearWgradArgs a = {};
    a.dtype = dtype; a.dy = dy; a.dy_is_f32 = dtype == MIVIT_F32; a.lddy = lddy;
    a.x = x; a.x_is_f32 = x_is_f32 || dtype == MIVIT_F32; a.ldx = ldx;
    a.M = M; a.N = N; a.K = K; a.dW = dW; a.db = db; a.accumulate = accumulate; a.ws = workspace; a.ws_bytes = workspace_bytes;
    return launch_linear_wgrad(a, static_cast<hipStream_t>(stream));
}
