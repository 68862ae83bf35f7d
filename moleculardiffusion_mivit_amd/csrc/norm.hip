// LayerNorm forward / backward (nn.LayerNorm over E, eps 1e-5; reference helpers/models.py:88-89,134,301).
// One 64-lane wavefront per row: the row lives in registers (E/64 values per lane), mean / variance are
// shuffle reductions, statistics are fp32 whatever the storage type.
#include "common.h"

namespace {

constexpr float LN_EPS = 1e-5f;

struct LnFwd {
    const void *z; int64_t ldz; const float *gamma; const float *beta;
    int M, E;
    void *y; int64_t ldy; int rows_per_seq, out_seq_stride, out_row_off;
    const float *pos; float *mean; float *rstd;
    int in_rows, in_stride, in_off;
};

__device__ __forceinline__ int64_t map_row(int r, int rows, int stride, int off) {
    return rows > 0 ? (int64_t)(r / rows) * stride + r % rows + off : (int64_t)r;
}

template <typename T, int EPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwd a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T *z = static_cast<const T *>(a.z);
    T *y = static_cast<T *>(a.y);
    const float invE = 1.f / (float)a.E;
    for (int r = blockIdx.x * 4 + wave; r < a.M; r += gridDim.x * 4) {
        const int64_t ir = map_row(r, a.in_rows, a.in_stride, a.in_off);
        const int64_t orow = map_row(r, a.rows_per_seq, a.out_seq_stride, a.out_row_off);
        float v[EPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < a.E ? to_f32(z[ir * a.ldz + c]) : 0.f;
            s += v[i];
        }
        const float mu = wave_sum(s) * invE;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int c = lane + 64 * i;
            const float d = c < a.E ? v[i] - mu : 0.f;
            q += d * d;
        }
        const float rs = rsqrtf(wave_sum(q) * invE + LN_EPS);
        const int p = a.rows_per_seq > 0 ? (int)(orow % a.out_seq_stride) : 0;
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int c = lane + 64 * i;
            if (c < a.E) {
                float o = (v[i] - mu) * rs * a.gamma[c] + a.beta[c];
                if (a.pos) o += a.pos[(int64_t)p * a.E + c];
                y[orow * a.ldy + c] = from_f32<T>(o);
            }
        }
        if (lane == 0) {
            if (a.mean) a.mean[r] = mu;
            if (a.rstd) a.rstd[r] = rs;
        }
    }
}

struct LnBwd {
    const void *dy; int64_t lddy; const void *z; int64_t ldz;
    const float *gamma; const float *mean; const float *rstd;
    int M, E;
    int rows_per_seq, in_seq_stride, in_row_off;
    int z_rows, z_stride, z_off;
    void *dz; int64_t lddz;
    float *part_g; float *part_b;   // [gridDim.x][E]
    float *part_z;                  // optional column sums of dz (vector path)
};

template <typename T, int EPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwd a) {
    __shared__ float red[2][4][64 * EPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T *dy = static_cast<const T *>(a.dy);
    const T *z = static_cast<const T *>(a.z);
    T *dz = static_cast<T *>(a.dz);
    const float invE = 1.f / (float)a.E;
    float gam[EPL], accg[EPL], accb[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int c = lane + 64 * i;
        gam[i] = c < a.E ? a.gamma[c] : 0.f;
        accg[i] = 0.f;
        accb[i] = 0.f;
    }
    for (int r = blockIdx.x * 4 + wave; r < a.M; r += gridDim.x * 4) {
        const int64_t dr = map_row(r, a.rows_per_seq, a.in_seq_stride, a.in_row_off);
        const int64_t zr = map_row(r, a.z_rows, a.z_stride, a.z_off);
        const bool hat = a.mean == nullptr;          // z already holds the normalised rows (fused layer blocks)
        const float mu = hat ? 0.f : a.mean[r], rs = a.rstd[r];
        float xh[EPL], g[EPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int c = lane + 64 * i;
            const bool ok = c < a.E;
            const float d = ok ? to_f32(dy[dr * a.lddy + c]) : 0.f;
            xh[i] = ok ? (hat ? to_f32(z[zr * a.ldz + c]) : (to_f32(z[zr * a.ldz + c]) - mu) * rs) : 0.f;
            g[i] = d * gam[i];
            accg[i] += d * xh[i];
            accb[i] += d;
            s1 += g[i];
            s2 += g[i] * xh[i];
        }
        s1 = wave_sum(s1) * invE;
        s2 = wave_sum(s2) * invE;
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int c = lane + 64 * i;
            if (c < a.E) dz[zr * a.lddz + c] = from_f32<T>(rs * (g[i] - s1 - xh[i] * s2));
        }
    }
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        red[0][wave][lane + 64 * i] = accg[i];
        red[1][wave][lane + 64 * i] = accb[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < a.E; c += 256) {
        a.part_g[(int64_t)blockIdx.x * a.E + c] = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        a.part_b[(int64_t)blockIdx.x * a.E + c] = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    }
}

// ---- vectorised fast path: 16-byte accesses, LPR lanes per row (64 / LPR rows per wavefront at once) ----
__device__ __forceinline__ float group_sum(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_vec(const LnFwd a, const int LPR) {
    constexpr int V = 16 / sizeof(T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int RPW = 64 / LPR, sub = lane / LPR, li = lane % LPR;
    const T *z = static_cast<const T *>(a.z);
    T *y = static_cast<T *>(a.y);
    const float invE = 1.f / (float)a.E;
    float gam[NV][V], bet[NV][V];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int c = (n * LPR + li) * V;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            gam[n][e] = c < a.E ? a.gamma[c + e] : 0.f;
            bet[n][e] = c < a.E ? a.beta[c + e] : 0.f;
        }
    }
    for (int r0 = (blockIdx.x * 4 + wave) * RPW; r0 < a.M; r0 += gridDim.x * 4 * RPW) {
        const int r = r0 + sub;
        const bool valid = r < a.M;
        const int64_t ir = map_row(valid ? r : 0, a.in_rows, a.in_stride, a.in_off);
        const int64_t orow = map_row(valid ? r : 0, a.rows_per_seq, a.out_seq_stride, a.out_row_off);
        float v[NV][V];
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int c = (n * LPR + li) * V;
            if (valid && c < a.E) load16(z + ir * a.ldz + c, v[n]);
            else
#pragma unroll
                for (int e = 0; e < V; ++e) v[n][e] = 0.f;
#pragma unroll
            for (int e = 0; e < V; ++e) s += v[n][e];
        }
        const float mu = group_sum(s, LPR) * invE;
        float q = 0.f;
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int c = (n * LPR + li) * V;
            if (c < a.E)
#pragma unroll
                for (int e = 0; e < V; ++e) { const float d = v[n][e] - mu; q += d * d; }
        }
        const float rs = rsqrtf(group_sum(q, LPR) * invE + LN_EPS);
        const int p = a.rows_per_seq > 0 ? (int)(orow % a.out_seq_stride) : 0;
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int c = (n * LPR + li) * V;
            if (valid && c < a.E) {
                float o[V];
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    o[e] = (v[n][e] - mu) * rs * gam[n][e] + bet[n][e];
                    if (a.pos) o[e] += a.pos[(int64_t)p * a.E + c + e];
                }
                store16(y + orow * a.ldy + c, o);
            }
        }
        if (li == 0 && valid) {
            if (a.mean) a.mean[r] = mu;
            if (a.rstd) a.rstd[r] = rs;
        }
    }
}

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_vec(const LnBwd a, const int LPR) {
    constexpr int V = 16 / sizeof(T);
    __shared__ float red[3][4][64 * NV * V];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int RPW = 64 / LPR, sub = lane / LPR, li = lane % LPR;
    const T *dy = static_cast<const T *>(a.dy);
    const T *z = static_cast<const T *>(a.z);
    T *dz = static_cast<T *>(a.dz);
    const float invE = 1.f / (float)a.E;
    float gam[NV][V], accg[NV][V], accb[NV][V], accz[NV][V];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int c = (n * LPR + li) * V;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            gam[n][e] = c < a.E ? a.gamma[c + e] : 0.f;
            accg[n][e] = accb[n][e] = accz[n][e] = 0.f;
        }
    }
    for (int r0 = (blockIdx.x * 4 + wave) * RPW; r0 < a.M; r0 += gridDim.x * 4 * RPW) {
        const int r = r0 + sub;
        const bool valid = r < a.M;
        const int64_t dr = map_row(valid ? r : 0, a.rows_per_seq, a.in_seq_stride, a.in_row_off);
        const int64_t zr = map_row(valid ? r : 0, a.z_rows, a.z_stride, a.z_off);
        const bool hat = a.mean == nullptr;          // z already holds the normalised rows (fused layer blocks)
        const float mu = (valid && !hat) ? a.mean[r] : 0.f, rs = valid ? a.rstd[r] : 0.f;
        float xh[NV][V], g[NV][V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int c = (n * LPR + li) * V;
            float d[V], zz[V];
            if (valid && c < a.E) {
                load16(dy + dr * a.lddy + c, d);
                load16(z + zr * a.ldz + c, zz);
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) { d[e] = 0.f; zz[e] = mu; }
            }
#pragma unroll
            for (int e = 0; e < V; ++e) {
                xh[n][e] = hat ? zz[e] : (zz[e] - mu) * rs;
                g[n][e] = d[e] * gam[n][e];
                accg[n][e] += d[e] * xh[n][e];
                accb[n][e] += d[e];
                s1 += g[n][e];
                s2 += g[n][e] * xh[n][e];
            }
        }
        s1 = group_sum(s1, LPR) * invE;
        s2 = group_sum(s2, LPR) * invE;
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int c = (n * LPR + li) * V;
            if (valid && c < a.E) {
                float o[V];
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    o[e] = rs * (g[n][e] - s1 - xh[n][e] * s2);
                    accz[n][e] += o[e];
                }
                store16(dz + zr * a.lddz + c, o);
            }
        }
    }
    // lanes with equal li hold the same columns: fold the 64/LPR row groups, then the 4 waves through LDS
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float x = accg[n][e], y2 = accb[n][e], w2 = accz[n][e];
            for (int o = LPR; o < 64; o <<= 1) {
                x += __shfl_xor(x, o, 64);
                y2 += __shfl_xor(y2, o, 64);
                w2 += __shfl_xor(w2, o, 64);
            }
            if (sub == 0) {
                const int c = (n * LPR + li) * V + e;
                red[0][wave][c] = x;
                red[1][wave][c] = y2;
                red[2][wave][c] = w2;
            }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < a.E; c += 256) {
        a.part_g[(int64_t)blockIdx.x * a.E + c] = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        a.part_b[(int64_t)blockIdx.x * a.E + c] = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        if (a.part_z) a.part_z[(int64_t)blockIdx.x * a.E + c] = red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c];
    }
}

struct VecPlan { bool ok; int NV, LPR; };
template <typename T>
VecPlan vec_plan(int E, int64_t ld_a, int64_t ld_b, const void *p0, const void *p1, const void *p2) {
    constexpr int V = 16 / sizeof(T);
    VecPlan v = {false, 1, 64};
    if (E % V || ld_a % V || ld_b % V) return v;
    for (const void *p : {p0, p1, p2})
        if (p && (reinterpret_cast<uintptr_t>(p) & 15)) return v;
    const int vecs = E / V;
    v.NV = vecs <= 64 ? 1 : (vecs <= 128 ? 2 : 4);
    if (vecs > 256) return v;
    const int need = (vecs + v.NV - 1) / v.NV;
    int l = 1;
    while (l < need) l <<= 1;
    v.LPR = l;
    v.ok = true;
    return v;
}

int ln_blocks(int M) {
    int b = ceil_div(M, 16);
    return b < 1 ? 1 : (b > 2048 ? 2048 : b);   // = per-block dgamma/dbeta partials to reduce afterwards (8 workgroups per CU)
}

template <typename T>
int ln_fwd_dispatch(const LnFwd &k, hipStream_t s) {
    const dim3 block(256);
    ProfScope prof(s);
    const VecPlan vp = vec_plan<T>(k.E, k.ldz, k.ldy, k.z, k.y, nullptr);
    if (vp.ok) {
        const int rows_per_block = 4 * (64 / vp.LPR);
        int blocks = ceil_div(k.M, rows_per_block);
        if (blocks > 2048) blocks = 2048;
        if (vp.NV == 1) hipLaunchKernelGGL((ln_fwd_vec<T, 1>), dim3(blocks), block, 0, s, k, vp.LPR);
        else if (vp.NV == 2) hipLaunchKernelGGL((ln_fwd_vec<T, 2>), dim3(blocks), block, 0, s, k, vp.LPR);
        else hipLaunchKernelGGL((ln_fwd_vec<T, 4>), dim3(blocks), block, 0, s, k, vp.LPR);
        MIVIT_LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid(ln_blocks(k.M));
    if (k.E <= 64) hipLaunchKernelGGL((ln_fwd_kernel<T, 1>), grid, block, 0, s, k);
    else if (k.E <= 128) hipLaunchKernelGGL((ln_fwd_kernel<T, 2>), grid, block, 0, s, k);
    else if (k.E <= 256) hipLaunchKernelGGL((ln_fwd_kernel<T, 4>), grid, block, 0, s, k);
    else if (k.E <= 512) hipLaunchKernelGGL((ln_fwd_kernel<T, 8>), grid, block, 0, s, k);
    else hipLaunchKernelGGL((ln_fwd_kernel<T, 16>), grid, block, 0, s, k);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int ln_bwd_dispatch(const LnBwd &k, int blocks, bool *did_z, hipStream_t s) {
    const dim3 grid(blocks), block(256);
    ProfScope prof(s);
    const VecPlan vp = vec_plan<T>(k.E, k.lddy, k.ldz, k.dy, k.z, k.dz);
    *did_z = false;
    if (vp.ok && k.lddz % (16 / (int)sizeof(T)) == 0) {
        *did_z = k.part_z != nullptr;
        if (vp.NV == 1) hipLaunchKernelGGL((ln_bwd_vec<T, 1>), grid, block, 0, s, k, vp.LPR);
        else if (vp.NV == 2) hipLaunchKernelGGL((ln_bwd_vec<T, 2>), grid, block, 0, s, k, vp.LPR);
        else hipLaunchKernelGGL((ln_bwd_vec<T, 4>), grid, block, 0, s, k, vp.LPR);
        MIVIT_LAUNCH_CHECK();
        return 0;
    }
    if (k.E <= 64) hipLaunchKernelGGL((ln_bwd_kernel<T, 1>), grid, block, 0, s, k);
    else if (k.E <= 128) hipLaunchKernelGGL((ln_bwd_kernel<T, 2>), grid, block, 0, s, k);
    else if (k.E <= 256) hipLaunchKernelGGL((ln_bwd_kernel<T, 4>), grid, block, 0, s, k);
    else if (k.E <= 512) hipLaunchKernelGGL((ln_bwd_kernel<T, 8>), grid, block, 0, s, k);
    else hipLaunchKernelGGL((ln_bwd_kernel<T, 16>), grid, block, 0, s, k);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int launch_layernorm_fwd(const LayerNormFwdArgs &a, hipStream_t s) {
    MIVIT_CHECK(a.M > 0 && a.E > 0 && a.E <= 1024, "layernorm_fwd: unsupported shape M=%d E=%d (E <= 1024)", a.M, a.E);
    LnFwd k = {a.z, a.ldz, a.gamma, a.beta, a.M, a.E, a.y, a.ldy, a.rows_per_seq, a.out_seq_stride, a.out_row_off,
               a.pos, a.mean, a.rstd, a.in_rows, a.in_stride, a.in_off};
    return a.dtype == MIVIT_F32 ? ln_fwd_dispatch<float>(k, s) : a.dtype == MIVIT_BF16 ? ln_fwd_dispatch<bf16>(k, s) : ln_fwd_dispatch<f16>(k, s);
}

size_t layernorm_bwd_ws_bytes(int M, int E) { return align_up((size_t)3 * ln_blocks(M) * E * sizeof(float), 256); }

int launch_layernorm_bwd(const LayerNormBwdArgs &a, hipStream_t s) {
    MIVIT_CHECK(a.M > 0 && a.E > 0 && a.E <= 1024, "layernorm_bwd: unsupported shape M=%d E=%d (E <= 1024)", a.M, a.E);
    MIVIT_CHECK(a.ws_bytes >= layernorm_bwd_ws_bytes(a.M, a.E), "layernorm_bwd: workspace too small");
    const int blocks = ln_blocks(a.M);
    float *pg = static_cast<float *>(a.ws), *pb = pg + (size_t)blocks * a.E, *pz = pb + (size_t)blocks * a.E;
    LnBwd k = {a.dy, a.lddy, a.z, a.ldz, a.gamma, a.mean, a.rstd, a.M, a.E, a.rows_per_seq, a.in_seq_stride,
               a.in_row_off, a.z_rows, a.z_stride, a.z_off, a.dz, a.lddz, pg, pb, a.dzsum ? pz : nullptr};
    bool did_z = false;
    int rc = a.dtype == MIVIT_F32 ? ln_bwd_dispatch<float>(k, blocks, &did_z, s)
           : a.dtype == MIVIT_BF16 ? ln_bwd_dispatch<bf16>(k, blocks, &did_z, s) : ln_bwd_dispatch<f16>(k, blocks, &did_z, s);
    if (rc) return rc;
    if (a.dgamma && a.dbeta) {
        const bool z = a.dzsum && did_z;
        if ((rc = launch_slab_reduce3(pg, a.dgamma, pb, a.dbeta, z ? pz : nullptr, z ? a.dzsum : nullptr, blocks, a.E,
                                      a.accumulate, s)))
            return rc;
    } else {
        if (a.dgamma && (rc = launch_slab_reduce(pg, blocks, a.E, a.dgamma, a.accumulate, s))) return rc;
        if (a.dbeta && (rc = launch_slab_reduce(pb, blocks, a.E, a.dbeta, a.accumulate, s))) return rc;
        if (a.dzsum && did_z && (rc = launch_slab_reduce(pz, blocks, a.E, a.dzsum, a.accumulate, s))) return rc;
    }
    if (a.dzsum && !did_z) return 2;   // caller must fall back to a column-sum kernel (scalar LayerNorm path was taken)
    return 0;
}

extern "C" int mivit_layernorm_fwd(int dtype, const void *z, int64_t ldz, const float *gamma, const float *beta, int M,
                                   int E, void *y, int64_t ldy, int rows_per_seq, int out_seq_stride, int out_row_off,
                                   const float *pos, float *mean, float *rstd, void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16 || dtype == MIVIT_F16, "bad dtype %d", dtype);
    MIVIT_CHECK(z && gamma && beta && y, "layernorm_fwd: null pointer");
    LayerNormFwdArgs a = {};
    a.dtype = dtype; a.z = z; a.ldz = ldz; a.gamma = gamma; a.beta = beta; a.M = M; a.E = E; a.y = y; a.ldy = ldy;
    a.rows_per_seq = rows_per_seq; a.out_seq_stride = out_seq_stride; a.out_row_off = out_row_off;
    a.pos = pos; a.mean = mean; a.rstd = rstd;
    return launch_layernorm_fwd(a, static_cast<hipStream_t>(stream));
}

extern "C" size_t mivit_layernorm_bwd_workspace_bytes(int M, int E) { return layernorm_bwd_ws_bytes(M, E); }

extern "C" int mivit_layernorm_bwd(int dtype, const void *dy, int64_t lddy, const void *z, int64_t ldz,
                                   const float *gamma, const float *mean, const float *rstd, int M, int E,
                                   int rows_per_seq, int in_seq_stride, int in_row_off, void *dz, int64_t lddz,
                                   float *dgamma, float *dbeta, int accumulate, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    MIVIT_CHECK(dtype == MIVIT_F32 || dtype == MIVIT_BF16 || dtype == MIVIT_F16, "bad dtype %d", dtype);
    MIVIT_CHECK(dy && z && gamma && mean && rstd && dz && workspace, "layernorm_bwd: null pointer");
    LayerNormBwdArgs a = {};
    a.dtype = dtype; a.dy = dy; a.lddy = lddy; a.z = z; a.ldz = ldz; a.gamma = gamma; a.mean = mean; a.rstd = rstd;
    a.M = M; a.E = E; a.rows_per_seq = rows_per_seq; a.in_seq_stride = in_seq_stride; a.in_row_off = in_row_off;
    a.dz = dz; a.lddz = lddz; a.dgamma = dgamma; a.dbeta = dbeta; a.accumulate = accumulate;
    a.ws = workspace; a.ws_bytes = workspace_bytes;
    return launch_layernorm_bwd(a, static_cast<hipStream_t>(stream));
}
