// Model-level engine: GeneralTransformer.forward and its backward as a fixed sequence of HIP launches
// (reference helpers/models.py:328-361 orchestration, :136-141 Transformer, :97-108 post-norm encoder layer,
// :33-59 attention, :72-77 feed-forward, :268-276 MLP head).  Host code only: every arithmetic step is one of
// the kernels in gemm.hip / norm.hip / attention.hip / misc.hip.  The engine owns no device memory: the caller
// passes the parameter arena, the gradient arena and one workspace whose layout is computed here.
#include "common.h"
#include <atomic>
#include <algorithm>

#include <string>
#include <vector>

// fused encoder-layer blocks (fused_fwd.hip / fused_bwd.hip): one set of externals per element type and layer width (elem.h:
// the two units are compiled as is, with -DMIVIT_ELEM_F16, with -DMIVIT_WIDTH64 and with both)
#define MIVIT_FUSED_DECLS(SFX)                                                                                                     \
    bool fused_layer_supported##SFX(int dtype, int E, int F, int H, int S);                                                        \
    int launch_attn_block_fwd##SFX(const void *nin, const float *gin, const float *bin, const void *Wqkv, const float *bqkv,       \
                                   const void *Wo, const float *bo, const float *gout, const float *bout, int B, int S, void *ctx, \
                                   void *nout, float *rstd, void *xout, void *zout, float *mean, void *qkvout, hipStream_t s);     \
    int launch_mlp_block_fwd##SFX(const void *nin, const float *gin, const float *bin, const void *W1, const float *b1,            \
                                  const void *W2, const float *b2, const float *gout, const float *bout, int M, int act,           \
                                  void *nout, float *rstd, void *xout, void *zout, float *mean, void *hout, void *uout,            \
                                  hipStream_t s);                                                                                  \
    size_t mlp_block_bwd_ws_bytes##SFX(int M);                                                                                     \
    size_t attn_out_bwd_ws_bytes##SFX(int M);                                                                                      \
    int launch_attn_out_bwd##SFX(const void *dy, const void *n1, const float *rstd1, const float *gamma1, const void *ctx,         \
                                 const void *Wo, int M, void *dz1, void *dctx, float *dWo, float *dbo, float *dgamma1,             \
                                 float *dbeta1, void *ws, size_t ws_bytes, hipStream_t s);                                         \
    int launch_mlp_block_bwd##SFX(const void *dy, const void *n2, const float *rstd2, const float *gamma2, const void *n1,         \
                                  const float *gamma1, const float *beta1, const void *W1, const float *b1, const void *W2, int M, \
                                  int act, void *dx1, float *dW1, float *db1, float *dW2, float *db2, float *dgamma2,              \
                                  float *dbeta2, void *ws, size_t ws_bytes, hipStream_t s);                                        \
    size_t qkv_bwd_ws_bytes##SFX(int M);                                                                                           \
    int launch_qkv_bwd##SFX(const void *dqkv, const void *x, const void *Wqkv, const void *res, int M, void *dx, float *dW,        \
                            float *db, const float *fix_gamma, const float *fix_beta, void *ws, size_t ws_bytes, hipStream_t s);
MIVIT_FUSED_DECLS()
MIVIT_FUSED_DECLS(_f16)
MIVIT_FUSED_DECLS(_w64)
MIVIT_FUSED_DECLS(_w64_f16)
#undef MIVIT_FUSED_DECLS

struct FusedOps {
    decltype(&fused_layer_supported) ok;
    decltype(&launch_attn_block_fwd) attn_fwd;
    decltype(&launch_mlp_block_fwd) mlp_fwd;
    decltype(&launch_mlp_block_bwd) mlp_bwd;
    decltype(&launch_attn_out_bwd) attn_out_bwd;
    decltype(&mlp_block_bwd_ws_bytes) mlp_bwd_ws;
    decltype(&attn_out_bwd_ws_bytes) attn_out_bwd_ws;
    decltype(&launch_qkv_bwd) qkv_bwd;
    decltype(&qkv_bwd_ws_bytes) qkv_bwd_ws;
};
#define MIVIT_FUSED_TABLE(SFX)                                                                                                \
    {fused_layer_supported##SFX, launch_attn_block_fwd##SFX, launch_mlp_block_fwd##SFX, launch_mlp_block_bwd##SFX,            \
     launch_attn_out_bwd##SFX, mlp_block_bwd_ws_bytes##SFX, attn_out_bwd_ws_bytes##SFX, launch_qkv_bwd##SFX, qkv_bwd_ws_bytes##SFX}
static const FusedOps kFusedBf16 = MIVIT_FUSED_TABLE(), kFusedF16 = MIVIT_FUSED_TABLE(_f16), kFusedBf16W64 = MIVIT_FUSED_TABLE(_w64),
                      kFusedF16W64 = MIVIT_FUSED_TABLE(_w64_f16);
#undef MIVIT_FUSED_TABLE
// (MIVIT_NO_FUSED_W64: the reference's shipped width back on the per-operator streaming path -- A/B measurements)
static const FusedOps *fused_ops(int dtype, int E) {
    static const bool f16_off = getenv("MIVIT_NO_F16_STREAM") != nullptr, w64_off = getenv("MIVIT_NO_FUSED_W64") != nullptr;
    if (E == 64 && w64_off) return nullptr;
    if (dtype == MIVIT_BF16) return E == 64 ? &kFusedBf16W64 : &kFusedBf16;
    if (dtype == MIVIT_F16 && !f16_off) return E == 64 ? &kFusedF16W64 : &kFusedF16;
    return nullptr;
}
static bool fused_ok(int dtype, int E, int F, int H, int S) {
    const FusedOps *f = fused_ops(dtype, E);
    return f && f->ok(dtype, E, F, H, S);
}

struct ParamInfo {
    std::string name;
    int64_t offset, numel;
};

struct LayerParams {
    int64_t qkv_w, qkv_b, out_w, out_b, n1_w, n1_b, fc1_w, fc1_b, fc2_w, fc2_b, n2_w, n2_b;
};

struct mivit_plan {
    mivit_config c;
    std::vector<ParamInfo> params;
    std::vector<std::pair<int64_t, int64_t>> stages;
    int64_t arena;
    // offsets (floats) into the arena
    int64_t tn_w, tn_b, fp0_w, fp0_b, fp2_w, fp2_b, h0_w, h0_b, h3_w, h3_b;
    std::vector<LayerParams> layers;
    int64_t reg, pos, n0_w, n0_b, emb_w, emb_b;
    int head_in;
    uint64_t uid;       // unique per created plan: hipGraph keys use it, never the pointer (a freed plan's address can be reused)
};

namespace {

constexpr int MAX_TOKENS = 128;   // helpers/models.py:8

int64_t add_param(mivit_plan *p, const std::string &name, int64_t numel, bool align = true) {
    // 32-byte aligned fp32 tensors = 16-byte aligned in the bf16 shadow copy (k/v follow q unpadded: one operand)
    if (align) p->arena = (p->arena + 7) / 8 * 8;
    const int64_t off = p->arena;
    p->params.push_back({name, off, numel});
    p->arena += numel;
    return off;
}

void add_feature_projector(mivit_plan *p) {
    const int E = p->c.embed_dim, G = p->c.global_feature_dim;
    p->fp0_w = add_param(p, "feature_projector.0.weight", (int64_t)E * G);
    p->fp0_b = add_param(p, "feature_projector.0.bias", E);
    p->fp2_w = add_param(p, "feature_projector.2.weight", (int64_t)E * E);
    p->fp2_b = add_param(p, "feature_projector.2.bias", E);
}

struct Ws {
    size_t total;
    size_t wsh;          // bf16 shadow of the parameter arena (bf16 mode), refreshed by every forward
    size_t emb, mean0, rstd0, x0;
    struct L { size_t qkv, ctx, z1, x1, h, u, z2, x2, mean1, rstd1, mean2, rstd2; };
    std::vector<L> layer;
    size_t xF, meanF, rstdF, pooled, fp_h, fp_out, head_in, hh;
    size_t xL;           // fused layer blocks: x = gamma * xhat + beta of the LAST layer (input of the final norm)
    // backward temporaries
    size_t dout_t, d_hh, d_head_in, d_pool_c, d_fp_h, dxa, dxb, dF, dctx, dqkv, wgrad, ln, colsum;
    size_t wgrad_bytes, ln_bytes, colsum_bytes;
};

Ws make_ws(const mivit_plan *p, int B, int T, bool bwd) {
    const mivit_config &c = p->c;
    const size_t ts = dtype_size(c.dtype);
    const int E = c.embed_dim, F = c.hidden_dim, L = c.num_layers;
    const int S = T + (c.use_regression_token ? 1 : 0);
    const size_t M = (size_t)B * S, Mt = (size_t)B * T;
    Ws w = {};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    w.wsh = c.dtype != MIVIT_F32 ? take((size_t)p->arena * 2) : 0;
    w.emb = take(Mt * E * ts); w.mean0 = take(Mt * 4); w.rstd0 = take(Mt * 4);
    w.x0 = take(M * E * ts);
    const int nsets = bwd ? L : 1;
    const bool fused = L > 0 && fused_ok(c.dtype, E, F, c.num_heads, S);
    for (int l = 0; l < nsets; ++l) {
        Ws::L s;
        if (fused) {
            // fused layer blocks: z1 / z2 hold the NORMALISED sub-layer outputs (xhat, bf16), rstd1 / rstd2 their 1/std;
            // q|k|v and h are kept only for the backward; no pre-norm sums, no means, no x copies
            s.qkv = bwd ? take(M * 3 * E * ts) : 0; s.ctx = take(M * E * ts); s.z1 = take(M * E * ts); s.x1 = 0;
            s.h = 0; s.u = 0;
            s.z2 = take(M * E * ts); s.x2 = 0;
            s.mean1 = s.mean2 = 0; s.rstd1 = take(M * 4); s.rstd2 = take(M * 4);
        } else {
            s.qkv = take(M * 3 * E * ts); s.ctx = take(M * E * ts); s.z1 = take(M * E * ts); s.x1 = take(M * E * ts);
            s.h = take(M * F * ts); s.u = c.activation == MIVIT_ACT_GELU ? take(M * F * ts) : 0;
            s.z2 = take(M * E * ts); s.x2 = take(M * E * ts);
            s.mean1 = take(M * 4); s.rstd1 = take(M * 4); s.mean2 = take(M * 4); s.rstd2 = take(M * 4);
        }
        w.layer.push_back(s);
    }
    for (int l = nsets; l < L; ++l) w.layer.push_back(w.layer[0]);
    w.xL = fused ? take(M * E * ts) : 0;
    w.xF = c.use_regression_token ? 0 : take(M * E * ts);
    w.meanF = take(M * 4); w.rstdF = take(M * 4);
    w.pooled = take((size_t)B * E * ts);
    w.fp_h = take((size_t)B * E * ts); w.fp_out = take((size_t)B * E * ts);
    w.head_in = take((size_t)B * 2 * E * ts);
    w.hh = take((size_t)B * c.head_hidden * ts);
    if (bwd) {
        w.dout_t = take((size_t)B * c.output_dim * ts);
        w.d_hh = take((size_t)B * c.head_hidden * ts);
        w.d_head_in = take((size_t)B * 2 * E * ts);
        w.d_pool_c = take((size_t)B * E * ts);
        w.d_fp_h = take((size_t)B * E * ts);
        w.dxa = take(M * E * ts); w.dxb = take(M * E * ts);
        w.dF = take(M * F * ts); w.dctx = take(M * E * ts); w.dqkv = take(M * 3 * E * ts);
        size_t wg = 0;
        auto mx = [&](int m, int n, int k) {
            size_t b = linear_wgrad_ws_bytes(m, n, k);
            // (the bf16 and f16 builds of the weight-gradient kernels size their workspaces identically)
            if (c.dtype != MIVIT_F32 && n % 128 == 0 && k % 128 == 0 && m >= 256) b += wgrad_dma_ws_bytes(m, n, k);
            if (c.dtype != MIVIT_F32 && m >= 256) b = std::max(b, wgrad_small_ws_bytes(m, n, k));
            if (c.dtype != MIVIT_F32 && m >= 256) b = std::max(b, embed_small_wgrad_ws_bytes(m, n, k));      // (the embedding's shape only)
            if (b > wg) wg = b;
        };
        mx((int)M, 3 * E, E); mx((int)M, E, E); mx((int)M, F, E); mx((int)M, E, F);
        if (fused) {
            const FusedOps *fo = fused_ops(c.dtype, E);
            // (one region each: their slab reductions are deferred to one launch per layer, so the three slab sets coexist)
            wg = std::max(wg, fo->mlp_bwd_ws((int)M) + fo->attn_out_bwd_ws((int)M) + fo->qkv_bwd_ws((int)M));
        }
        if (c.embedding != MIVIT_EMBED_EXTERNAL) {
            mx((int)Mt, E, c.patch_size * c.patch_size);
            if ((c.dtype == MIVIT_F16 ? embed_dma_supported_f16 : embed_dma_supported)(c.dtype, (int)Mt, c.patch_size * c.patch_size, E)) {
                const size_t b = embed_wgrad_dma_ws_bytes((int)Mt, c.patch_size * c.patch_size, E);      // (same for both element types)
                if (b > wg) wg = b;
            }
        }
        mx(B, c.head_hidden, p->head_in); mx(B, c.output_dim, c.head_hidden);
        if (c.fusion != MIVIT_FUSION_NONE) { mx(B, E, E); mx(B, E, c.global_feature_dim); }
        w.wgrad_bytes = wg; w.wgrad = take(wg);
        w.ln_bytes = layernorm_bwd_ws_bytes((int)M, E); w.ln = take(w.ln_bytes);
        w.colsum_bytes = batch_colsum_ws_bytes(B, S, E); w.colsum = take(w.colsum_bytes);
    }
    w.total = off;
    return w;
}

inline void *at(void *base, size_t off) { return static_cast<char *>(base) + off; }
inline const void *at(const void *base, size_t off) { return static_cast<const char *>(base) + off; }
// pointer `cols` elements into a row of a T matrix
inline void *col_ptr(void *p, size_t cols, int dtype) { return static_cast<char *>(p) + cols * dtype_size(dtype); }

#define RC(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

// The streaming kernels exist once per 16-bit element type (elem.h: each unit is compiled for bf16 and, with
// -DMIVIT_ELEM_F16, for IEEE half): the plan's dtype picks the set.  fp32 (parity mode) has none: general kernels.
struct StreamOps {
    decltype(&rowstream_supported) rowstream_ok;
    decltype(&wavestream_supported) wavestream_ok;
    decltype(&launch_rowstream) rowstream;
    decltype(&wgrad_dma_supported) wgrad_dma_ok;
    decltype(&wgrad_dma_ws_bytes) wgrad_dma_ws;
    decltype(&launch_wgrad_dma) wgrad_dma;
    decltype(&wgrad_small_supported) wgrad_small_ok;
    decltype(&wgrad_small_ws_bytes) wgrad_small_ws;
    decltype(&launch_wgrad_small) wgrad_small;
    decltype(&embed_dma_supported) embed_ok;
    decltype(&launch_embed_fwd_dma) embed_fwd;
    decltype(&embed_wgrad_dma_ws_bytes) embed_wgrad_ws;
    decltype(&launch_embed_wgrad_dma) embed_wgrad;
    decltype(&embed_small_fwd_supported) embed_small_fwd_ok;          // small frames (any row length <= 256 pixels)
    decltype(&launch_embed_small_fwd) embed_small_fwd;
    decltype(&embed_small_wgrad_supported) embed_small_wgrad_ok;
    decltype(&embed_small_wgrad_ws_bytes) embed_small_wgrad_ws;
    decltype(&launch_embed_small_wgrad) embed_small_wgrad;
};
static const StreamOps kStreamBf16 = {rowstream_supported, wavestream_supported, launch_rowstream, wgrad_dma_supported, wgrad_dma_ws_bytes,
                                      launch_wgrad_dma, wgrad_small_supported, wgrad_small_ws_bytes, launch_wgrad_small,
                                      embed_dma_supported, launch_embed_fwd_dma, embed_wgrad_dma_ws_bytes, launch_embed_wgrad_dma,
                                      embed_small_fwd_supported, launch_embed_small_fwd, embed_small_wgrad_supported,
                                      embed_small_wgrad_ws_bytes, launch_embed_small_wgrad};
static const StreamOps kStreamF16 = {rowstream_supported_f16, wavestream_supported_f16, launch_rowstream_f16, wgrad_dma_supported_f16,
                                     wgrad_dma_ws_bytes_f16, launch_wgrad_dma_f16, wgrad_small_supported_f16, wgrad_small_ws_bytes_f16,
                                     launch_wgrad_small_f16, embed_dma_supported_f16, launch_embed_fwd_dma_f16,
                                     embed_wgrad_dma_ws_bytes_f16, launch_embed_wgrad_dma_f16, embed_small_fwd_supported_f16,
                                     launch_embed_small_fwd_f16, embed_small_wgrad_supported_f16, embed_small_wgrad_ws_bytes_f16,
                                     launch_embed_small_wgrad_f16};
static const StreamOps *stream_ops(int dtype) {
    static const bool f16_off = getenv("MIVIT_NO_F16_STREAM") != nullptr;        // (A/B: fp16 on the general kernels, as in rounds 1-2)
    return dtype == MIVIT_BF16 ? &kStreamBf16 : (dtype == MIVIT_F16 && !f16_off ? &kStreamF16 : nullptr);
}
// row-stream (DMA ring) or wave-stream kernels: launch_rowstream picks between the two families
static bool stream_gemm_supported(const StreamOps *so, int M, int N, int K, bool dgrad, int64_t lda, int64_t ldw, const void *A, const void *W) {
    return so && (so->rowstream_ok(M, N, K, dgrad, lda, ldw, A, W) || so->wavestream_ok(M, N, K, dgrad, lda, ldw, A, W));
}

int lin_fwd(int dtype, const void *x, int x_f32, int64_t ldx, const void *W, const float *b, int M, int N, int K,
            int act, const void *resid, int64_t ldr, void *y, int64_t ldy, void *pre, int y_f32, hipStream_t s) {
    const StreamOps *so = stream_ops(dtype);
    if (!x_f32 && !y_f32 && stream_gemm_supported(so, M, N, K, false, ldx, K, x, W) &&
        (!resid || ldr % 8 == 0) && ldy % 8 == 0) {
        prof_set_tag(MIVIT_PROF_LINEAR_FWD);
        return so->rowstream(false, x, ldx, W, K, M, N, K, b, act, nullptr, 0, 0, resid, ldr, y, ldy, pre, nullptr,
                                nullptr, nullptr, 0, nullptr, nullptr, s);
    }
    if (dtype == MIVIT_BF16 && !x_f32 && !y_f32 && gemm_dma_supported(M, N, K, false) && ldx % 8 == 0 && ldy % 8 == 0 &&
        (!resid || ldr % 8 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(y) |
                                      reinterpret_cast<uintptr_t>(resid) | reinterpret_cast<uintptr_t>(pre)) & 15) == 0) {
        prof_set_tag(MIVIT_PROF_LINEAR_FWD);
        return launch_gemm_dma_fwd(x, ldx, W, b, M, N, K, act, resid, ldr, y, ldy, pre, s);
    }
    if (so && x_f32 && !y_f32 && ldx == K && ldy == N && act == MIVIT_ACT_NONE && !resid && !pre &&
        so->embed_small_fwd_ok(M, K, N, x, y)) {          // the linear embedding of small frames (fp32 rows of any length)
        prof_set_tag(MIVIT_PROF_EMBED_FWD);
        return so->embed_small_fwd(static_cast<const float *>(x), W, b, y, M, K, N, s);
    }
    LinearFwdArgs a = {};
    a.dtype = dtype; a.x = x; a.x_is_f32 = x_f32 || dtype == MIVIT_F32; a.ldx = ldx; a.W = W;
    a.w_is_bf16 = dtype != MIVIT_F32; a.bias = b;
    a.M = M; a.N = N; a.K = K; a.act = act; a.resid = resid; a.ldr = ldr; a.y = y; a.ldy = ldy; a.y_preact = pre;
    a.y_is_f32 = y_f32;
    prof_set_tag(x_f32 && K > 1024 ? MIVIT_PROF_EMBED_FWD : MIVIT_PROF_LINEAR_FWD);
    return launch_linear_fwd(a, s);
}
int lin_dgrad(int dtype, const void *dy, int64_t lddy, const void *W, int M, int N, int K, int act, const void *saved,
              int64_t lds, const void *dres, int64_t lddr, void *dx, int64_t lddx, int dx_f32, hipStream_t s) {
    const StreamOps *so = stream_ops(dtype);
    if (!dx_f32 && stream_gemm_supported(so, M, K, N, true, lddy, K, dy, W) && lddx % 8 == 0 &&
        (!dres || lddr % 8 == 0) && (act == MIVIT_ACT_NONE || lds % 8 == 0)) {
        prof_set_tag(MIVIT_PROF_LINEAR_DGRAD);
        return so->rowstream(true, dy, lddy, W, K, M, K, N, nullptr, MIVIT_ACT_NONE, act != MIVIT_ACT_NONE ? saved : nullptr,
                                lds, act, dres, lddr, dx, lddx, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, s);
    }
    if (dtype == MIVIT_BF16 && !dx_f32 && gemm_dma_supported(M, K, N, true) && lddy % 8 == 0 && lddx % 8 == 0 &&
        (!dres || lddr % 8 == 0) && (act == MIVIT_ACT_NONE || lds % 8 == 0) &&
        ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(dx) |
          reinterpret_cast<uintptr_t>(dres) | reinterpret_cast<uintptr_t>(act != MIVIT_ACT_NONE ? saved : nullptr)) & 15) == 0) {
        prof_set_tag(MIVIT_PROF_LINEAR_DGRAD);
        return launch_gemm_dma_dgrad(dy, lddy, W, M, N, K, act, saved, lds, dres, lddr, dx, lddx, s);
    }
    LinearDgradArgs a = {};
    a.dtype = dtype; a.dy = dy; a.dy_is_f32 = dtype == MIVIT_F32; a.lddy = lddy; a.W = W;
    a.w_is_bf16 = dtype != MIVIT_F32; a.M = M; a.N = N; a.K = K;
    a.act = act; a.saved = saved; a.lds = lds; a.dres = dres; a.lddr = lddr; a.dx = dx; a.lddx = lddx; a.dx_is_f32 = dx_f32;
    prof_set_tag(MIVIT_PROF_LINEAR_DGRAD);
    return launch_linear_dgrad(a, s);
}
int lin_wgrad(int dtype, const void *dy, int64_t lddy, const void *x, int x_f32, int64_t ldx, int M, int N, int K,
              float *dW, float *db, void *ws, size_t wsb, hipStream_t s) {
    const StreamOps *so = stream_ops(dtype);
    if (so && !x_f32 && dW && so->wgrad_dma_ok(M, N, K, lddy, ldx, dy, x) &&
        wsb >= so->wgrad_dma_ws(M, N, K) + linear_wgrad_ws_bytes(M, N, K)) {
        prof_set_tag(MIVIT_PROF_LINEAR_WGRAD);
        return so->wgrad_dma(dy, lddy, x, ldx, M, N, K, dW, db, ws, wsb, s);      // db (optional) from the same pass
    }
    if (so && !x_f32 && dW && so->wgrad_small_ok(M, N, K, lddy, ldx, dy, x) &&
        wsb >= so->wgrad_small_ws(M, N, K) && wsb >= linear_wgrad_ws_bytes(M, N, K)) {
        prof_set_tag(MIVIT_PROF_LINEAR_WGRAD);
        return so->wgrad_small(dy, lddy, x, ldx, M, N, K, dW, db, ws, wsb, s);      // db (optional) from the same pass
    }
    if (so && x_f32 && dW && ldx == K && so->embed_small_wgrad_ok(M, N, K, lddy, dy, x) && wsb >= so->embed_small_wgrad_ws(M, N, K)) {
        prof_set_tag(MIVIT_PROF_EMBED_WGRAD);
        return so->embed_small_wgrad(dy, lddy, static_cast<const float *>(x), M, N, K, dW, db, ws, wsb, s);
    }
    LinearWgradArgs a = {};
    a.dtype = dtype; a.dy = dy; a.dy_is_f32 = dtype == MIVIT_F32; a.lddy = lddy; a.x = x;
    a.x_is_f32 = x_f32 || dtype == MIVIT_F32; a.ldx = ldx; a.M = M; a.N = N; a.K = K; a.dW = dW; a.db = db;
    a.ws = ws; a.ws_bytes = wsb;
    prof_set_tag(x_f32 && K > 1024 ? MIVIT_PROF_EMBED_WGRAD : MIVIT_PROF_LINEAR_WGRAD);
    return launch_linear_wgrad(a, s);
}

// z = x W^T + b + resid;  y = LayerNorm(z)  (post-norm sub-layer, models.py:100-106): one row-stream launch when the
// block owns whole rows, otherwise GEMM + LayerNorm kernel.
int lin_res_ln(int dtype, const void *x, int64_t ldx, const void *W, const float *b, int M, int N, int K, const void *resid,
               void *z, const float *gamma, const float *beta, void *y, float *mean, float *rstd, hipStream_t s) {
    const StreamOps *so = stream_ops(dtype);
    if ((N == 128 || N == 64) && stream_gemm_supported(so, M, N, K, false, ldx, K, x, W)) {
        prof_set_tag(MIVIT_PROF_LINEAR_FWD);
        return so->rowstream(false, x, ldx, W, K, M, N, K, b, MIVIT_ACT_NONE, nullptr, 0, 0, resid, N, z, N, nullptr, gamma,
                                beta, y, N, mean, rstd, s);
    }
    RC(lin_fwd(dtype, x, 0, ldx, W, b, M, N, K, MIVIT_ACT_NONE, resid, N, z, N, nullptr, 0, s));
    LayerNormFwdArgs n = {};
    n.dtype = dtype; n.z = z; n.ldz = N; n.gamma = gamma; n.beta = beta; n.M = M; n.E = N; n.y = y; n.ldy = N;
    n.mean = mean; n.rstd = rstd;
    prof_set_tag(MIVIT_PROF_LN_FWD);
    return launch_layernorm_fwd(n, s);
}

// LayerNorm backward that also produces the bias gradient of the Linear feeding it (column sums of dz).
// *need_colsum is set when the scalar LayerNorm path ran and the caller must compute that bias gradient itself.
int ln_bwd_bias(LayerNormBwdArgs &a, float *db, bool *need_colsum, hipStream_t s) {
    a.dzsum = db;
    prof_set_tag(MIVIT_PROF_LN_BWD);
    const int rc = launch_layernorm_bwd(a, s);
    *need_colsum = (rc == 2);
    return rc == 2 ? 0 : rc;
}

int check_call(const mivit_plan *plan, int B, int T, size_t ws_bytes, bool bwd, const char *who) {
    MIVIT_CHECK(plan, "%s: null plan", who);
    MIVIT_CHECK(B > 0 && T > 0, "%s: empty batch (B=%d, T=%d)", who, B, T);
    const int S = T + (plan->c.use_regression_token ? 1 : 0);
    MIVIT_CHECK(!plan->c.use_pos_encoding || S <= MAX_TOKENS, "%s: %d tokens exceed the %d-entry positional table",
                who, S, MAX_TOKENS);
    const int Dh = plan->c.embed_dim / plan->c.num_heads;
    MIVIT_CHECK(S <= attention_max_seq(plan->c.dtype, Dh),
                "%s: sequence of %d tokens (head dim %d) does not fit the LDS-resident attention kernel (max %d)", who, S,
                Dh, attention_max_seq(plan->c.dtype, Dh));
    MIVIT_CHECK((int64_t)B * S * 3 * plan->c.embed_dim < (1ll << 31) && (int64_t)B * S * plan->c.hidden_dim < (1ll << 31),
                "%s: batch too large for 32-bit row indexing", who);
    const Ws w = make_ws(plan, B, T, bwd);
    MIVIT_CHECK(ws_bytes >= w.total, "%s: workspace too small (%zu < %zu bytes)", who, ws_bytes, w.total);
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
extern "C" mivit_plan *mivit_plan_create(const mivit_config *cfg) {
    if (!cfg) { mivit_set_error("plan_create: null config"); return nullptr; }
    const mivit_config &c = *cfg;
#define PLAN_CHECK(cond, ...) do { if (!(cond)) { mivit_set_error(__VA_ARGS__); return nullptr; } } while (0)
    PLAN_CHECK(c.abi_version == MIVIT_ABI_VERSION, "plan_create: ABI version %d != %d", c.abi_version, MIVIT_ABI_VERSION);
    PLAN_CHECK(c.dtype == MIVIT_F32 || c.dtype == MIVIT_BF16 || c.dtype == MIVIT_F16, "plan_create: bad dtype %d", c.dtype);
    PLAN_CHECK(c.embedding >= MIVIT_EMBED_LINEAR && c.embedding <= MIVIT_EMBED_EXTERNAL, "plan_create: bad embedding %d", c.embedding);
    PLAN_CHECK(c.embed_dim > 0 && c.num_heads > 0 && c.hidden_dim > 0 && c.num_layers >= 0, "plan_create: bad model dims");
    PLAN_CHECK(c.embed_dim % c.num_heads == 0, "embed_dim must be divisible by num_heads");
    PLAN_CHECK(c.embed_dim <= 1024, "plan_create: embed_dim %d > 1024 is not supported", c.embed_dim);
    PLAN_CHECK(c.embedding == MIVIT_EMBED_EXTERNAL || c.patch_size > 0, "plan_create: bad patch_size");
    PLAN_CHECK(c.activation >= MIVIT_ACT_RELU && c.activation <= MIVIT_ACT_GELU, "plan_create: unsupported activation %d", c.activation);
    PLAN_CHECK(c.fusion >= MIVIT_FUSION_NONE && c.fusion <= MIVIT_FUSION_LATE, "plan_create: bad fusion %d", c.fusion);
    PLAN_CHECK(c.fusion == MIVIT_FUSION_NONE || c.global_feature_dim > 0, "Must provide global_feature_dim if using global features");
    PLAN_CHECK(c.fusion != MIVIT_FUSION_EARLY || c.use_regression_token, "plan_create: early fusion needs the regression token");
    PLAN_CHECK(c.head_hidden > 0 && c.output_dim > 0, "plan_create: bad head dims");
#undef PLAN_CHECK
    mivit_plan *p = new mivit_plan();
    static std::atomic<uint64_t> next_uid{1};
    p->uid = next_uid.fetch_add(1);
    p->c = c;
    p->arena = 0;
    const int E = c.embed_dim, F = c.hidden_dim;
    p->head_in = c.fusion == MIVIT_FUSION_LATE ? 2 * E : E;
    // stage 0: final norm + (late-fusion feature projector) + head
    int64_t b0 = p->arena;
    p->tn_w = add_param(p, "transformer.norm.weight", E);
    p->tn_b = add_param(p, "transformer.norm.bias", E);
    if (c.fusion == MIVIT_FUSION_LATE) add_feature_projector(p);
    p->h0_w = add_param(p, "mlp_head.mlp.0.weight", (int64_t)c.head_hidden * p->head_in);
    p->h0_b = add_param(p, "mlp_head.mlp.0.bias", c.head_hidden);
    p->h3_w = add_param(p, "mlp_head.mlp.3.weight", (int64_t)c.output_dim * c.head_hidden);
    p->h3_b = add_param(p, "mlp_head.mlp.3.bias", c.output_dim);
    p->arena = (p->arena + 7) / 8 * 8;
    p->stages.push_back({b0, p->arena});
    // stages 1..L: encoder layers L-1 .. 0 (q/k/v weights and biases contiguous: one [3E,E] GEMM operand)
    p->layers.resize(c.num_layers);
    for (int l = c.num_layers - 1; l >= 0; --l) {
        b0 = p->arena;
        const std::string pre = "transformer.encoder_layers." + std::to_string(l) + ".";
        LayerParams &lp = p->layers[l];
        lp.qkv_w = add_param(p, pre + "self_attn.q_proj.weight", (int64_t)E * E);
        add_param(p, pre + "self_attn.k_proj.weight", (int64_t)E * E, false);
        add_param(p, pre + "self_attn.v_proj.weight", (int64_t)E * E, false);
        lp.qkv_b = add_param(p, pre + "self_attn.q_proj.bias", E);
        add_param(p, pre + "self_attn.k_proj.bias", E, false);
        add_param(p, pre + "self_attn.v_proj.bias", E, false);
        lp.out_w = add_param(p, pre + "self_attn.out_proj.weight", (int64_t)E * E);
        lp.out_b = add_param(p, pre + "self_attn.out_proj.bias", E);
        lp.n1_w = add_param(p, pre + "norm1.weight", E);
        lp.n1_b = add_param(p, pre + "norm1.bias", E);
        lp.fc1_w = add_param(p, pre + "feed_forward.fc1.weight", (int64_t)F * E);
        lp.fc1_b = add_param(p, pre + "feed_forward.fc1.bias", F);
        lp.fc2_w = add_param(p, pre + "feed_forward.fc2.weight", (int64_t)E * F);
        lp.fc2_b = add_param(p, pre + "feed_forward.fc2.bias", E);
        lp.n2_w = add_param(p, pre + "norm2.weight", E);
        lp.n2_b = add_param(p, pre + "norm2.bias", E);
        p->arena = (p->arena + 7) / 8 * 8;
        p->stages.push_back({b0, p->arena});
    }
    // last stage: token assembly + embedding (+ early-fusion feature projector)
    b0 = p->arena;
    p->reg = c.use_regression_token ? add_param(p, "reg_token", E) : -1;
    p->pos = c.use_pos_encoding ? add_param(p, "transformer.pos_embedding", (int64_t)MAX_TOKENS * E) : -1;
    p->n0_w = add_param(p, "norm.weight", E);
    p->n0_b = add_param(p, "norm.bias", E);
    if (c.fusion == MIVIT_FUSION_EARLY) add_feature_projector(p);
    p->emb_w = p->emb_b = -1;
    if (c.embedding == MIVIT_EMBED_LINEAR) {
        p->emb_w = add_param(p, "embedding.proj.weight", (int64_t)E * c.patch_size * c.patch_size);
        p->emb_b = add_param(p, "embedding.proj.bias", E);
    } else if (c.embedding == MIVIT_EMBED_CNN) {
        p->emb_w = add_param(p, "embedding.conv.weight", (int64_t)E * c.patch_size * c.patch_size);
        p->emb_b = add_param(p, "embedding.conv.bias", E);
    }
    p->arena = (p->arena + 7) / 8 * 8;
    p->stages.push_back({b0, p->arena});
    // q/k/v must be contiguous (E*E and E are multiples of 4 whenever E is): verify
    for (const LayerParams &lp : p->layers) (void)lp;
    if ((int64_t)E * E % 4 != 0 || E % 4 != 0) {
        mivit_set_error("plan_create: embed_dim must be a multiple of 4 (got %d)", E);
        delete p;
        return nullptr;
    }
    return p;
}

extern "C" void mivit_plan_destroy(mivit_plan *plan) { delete plan; }
extern "C" int mivit_plan_num_params(const mivit_plan *plan) { return plan ? (int)plan->params.size() : 0; }
extern "C" const char *mivit_plan_param_name(const mivit_plan *plan, int i) {
    return (plan && i >= 0 && i < (int)plan->params.size()) ? plan->params[i].name.c_str() : nullptr;
}
extern "C" int64_t mivit_plan_param_offset(const mivit_plan *plan, int i) {
    return (plan && i >= 0 && i < (int)plan->params.size()) ? plan->params[i].offset : -1;
}
extern "C" int64_t mivit_plan_param_numel(const mivit_plan *plan, int i) {
    return (plan && i >= 0 && i < (int)plan->params.size()) ? plan->params[i].numel : -1;
}
extern "C" int64_t mivit_plan_arena_numel(const mivit_plan *plan) { return plan ? plan->arena : 0; }
extern "C" int mivit_plan_num_stages(const mivit_plan *plan) { return plan ? (int)plan->stages.size() : 0; }
extern "C" int mivit_plan_stage_range(const mivit_plan *plan, int stage, int64_t *begin, int64_t *end) {
    MIVIT_CHECK(plan && stage >= 0 && stage < (int)plan->stages.size(), "stage_range: bad stage %d", stage);
    if (begin) *begin = plan->stages[stage].first;
    if (end) *end = plan->stages[stage].second;
    return 0;
}
extern "C" size_t mivit_plan_workspace_bytes(const mivit_plan *plan, int B, int T, int need_backward) {
    if (!plan || B <= 0 || T <= 0) return 0;
    return make_ws(plan, B, T, need_backward != 0).total;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
static int forward_impl(const mivit_plan *plan, const float *params, const float *x, const float *features, int B,
                             int T, void *workspace, size_t workspace_bytes, int need_backward, float *out,
                             void *stream) {
    RC(check_call(plan, B, T, workspace_bytes, need_backward != 0, "mivit_forward"));
    MIVIT_CHECK(params && x && workspace && out, "mivit_forward: null pointer");
    const mivit_config &c = plan->c;
    MIVIT_CHECK(c.fusion == MIVIT_FUSION_NONE || features, "Global features required for %s fusion",
                c.fusion == MIVIT_FUSION_EARLY ? "early" : "late");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int dt = c.dtype, E = c.embed_dim, F = c.hidden_dim, H = c.num_heads, Dh = E / H;
    const int off = c.use_regression_token ? 1 : 0, S = T + off, M = B * S, Mt = B * T;
    const Ws w = make_ws(plan, B, T, need_backward != 0);
    void *ws = workspace;
    const float *P = params;
    // bf16 mode: one conversion of the whole fp32 arena per step; every GEMM then stages bf16 weights
    if (dt != MIVIT_F32) RC(launch_convert(1, P, plan->arena, 0, at(ws, w.wsh), plan->arena, 1, (int)plan->arena, 0, s, dt));
    auto WT = [&](int64_t off) -> const void * {
        return dt != MIVIT_F32 ? static_cast<const void *>(static_cast<const bf16 *>(at(ws, w.wsh)) + off)
                                : static_cast<const void *>(P + off);
    };

    // 1. frame embedding: one token per whole frame (models.py:146-199), [B*T, P*P] x [E, P*P]^T
    if (c.embedding == MIVIT_EMBED_EXTERNAL) {
        RC(launch_convert(1, x, E, dt == MIVIT_F32, at(ws, w.emb), E, Mt, E, 0, s, dt));
    } else {
        const int K = c.patch_size * c.patch_size;
        const StreamOps *so = stream_ops(dt);
        if (so && so->embed_ok(dt, Mt, K, E)) {
            prof_set_tag(MIVIT_PROF_EMBED_FWD);
            RC(so->embed_fwd(x, WT(plan->emb_w), P + plan->emb_b, at(ws, w.emb), Mt, K, E, s));
        } else {
            RC(lin_fwd(dt, x, 1, K, WT(plan->emb_w), P + plan->emb_b, Mt, E, K, MIVIT_ACT_NONE, nullptr, 0, at(ws, w.emb), E,
                       nullptr, 0, s));
        }
    }
    // 2. LayerNorm of the tokens, written behind the regression-token row, + positional table (models.py:334,347,138)
    {
        LayerNormFwdArgs a = {};
        a.dtype = dt; a.z = at(ws, w.emb); a.ldz = E; a.gamma = P + plan->n0_w; a.beta = P + plan->n0_b; a.M = Mt; a.E = E;
        a.y = at(ws, w.x0); a.ldy = E; a.rows_per_seq = T; a.out_seq_stride = S; a.out_row_off = off;
        a.pos = c.use_pos_encoding ? P + plan->pos : nullptr;
        a.mean = static_cast<float *>(at(ws, w.mean0)); a.rstd = static_cast<float *>(at(ws, w.rstd0));
        prof_set_tag(MIVIT_PROF_LN_FWD); RC(launch_layernorm_fwd(a, s));
    }
    // feature projector (models.py:316-320): Linear(Fg,E) -> ReLU -> Linear(E,E)
    if (c.fusion != MIVIT_FUSION_NONE) {
        const int G = c.global_feature_dim;
        RC(lin_fwd(dt, features, 1, G, WT(plan->fp0_w), P + plan->fp0_b, B, E, G, MIVIT_ACT_RELU, nullptr, 0,
                   at(ws, w.fp_h), E, nullptr, 0, s));
        RC(lin_fwd(dt, at(ws, w.fp_h), 0, E, WT(plan->fp2_w), P + plan->fp2_b, B, E, E, MIVIT_ACT_NONE, nullptr, 0,
                   at(ws, w.fp_out), E, nullptr, 0, s));
    }
    // 3. regression token row (models.py:339-347)
    if (c.use_regression_token)
        RC(launch_reg_token_fill(dt, at(ws, w.x0), B, S, E, P + plan->reg,
                                 c.fusion == MIVIT_FUSION_EARLY ? at(ws, w.fp_out) : nullptr,
                                 c.use_pos_encoding ? P + plan->pos : nullptr, s));
    // 4. encoder layers (post-norm, models.py:97-108)
    const void *xin = at(ws, w.x0);
    const bool fused = c.num_layers > 0 && fused_ok(dt, E, F, H, S);
    if (fused) {
        // fused layer blocks (fused_fwd.hip): the layers hand each other NORMALISED tokens, the consumer applies the producing
        // LayerNorm's affine (folded into its weights); training keeps q|k|v and h for the backward kernels, inference
        // nothing; only the last block materialises x for the final norm
        const float *gin = nullptr, *bin = nullptr;
        const void *nin = xin;
        for (int l = 0; l < c.num_layers; ++l) {
            const LayerParams &lp = plan->layers[l];
            const Ws::L &b = w.layer[l];
            const bool last = l + 1 == c.num_layers;
            prof_set_tag(MIVIT_PROF_ATTN_BLOCK_FWD);
            RC(fused_ops(dt, E)->attn_fwd(nin, gin, bin, WT(lp.qkv_w), P + lp.qkv_b, WT(lp.out_w), P + lp.out_b, P + lp.n1_w,
                                     P + lp.n1_b, B, S, at(ws, b.ctx), at(ws, b.z1), static_cast<float *>(at(ws, b.rstd1)),
                                     nullptr, nullptr, nullptr, need_backward ? at(ws, b.qkv) : nullptr, s));
            prof_set_tag(MIVIT_PROF_MLP_BLOCK_FWD);
            RC(fused_ops(dt, E)->mlp_fwd(at(ws, b.z1), P + lp.n1_w, P + lp.n1_b, WT(lp.fc1_w), P + lp.fc1_b, WT(lp.fc2_w), P + lp.fc2_b,
                                    P + lp.n2_w, P + lp.n2_b, M, c.activation, at(ws, b.z2), static_cast<float *>(at(ws, b.rstd2)),
                                    last ? at(ws, w.xL) : nullptr, nullptr, nullptr, nullptr, nullptr, s));      // (h is recomputed by the fused backward)
            nin = at(ws, b.z2); gin = P + lp.n2_w; bin = P + lp.n2_b;
        }
        xin = at(ws, w.xL);
    } else
    for (int l = 0; l < c.num_layers; ++l) {
        const LayerParams &lp = plan->layers[l];
        const Ws::L &b = w.layer[l];
        RC(lin_fwd(dt, xin, 0, E, WT(lp.qkv_w), P + lp.qkv_b, M, 3 * E, E, MIVIT_ACT_NONE, nullptr, 0, at(ws, b.qkv),
                   3 * E, nullptr, 0, s));
        prof_set_tag(MIVIT_PROF_ATTN_FWD); RC(launch_attention_fwd(dt, at(ws, b.qkv), B, S, H, Dh, at(ws, b.ctx), s));
        RC(lin_res_ln(dt, at(ws, b.ctx), E, WT(lp.out_w), P + lp.out_b, M, E, E, xin, at(ws, b.z1), P + lp.n1_w, P + lp.n1_b,
                      at(ws, b.x1), static_cast<float *>(at(ws, b.mean1)), static_cast<float *>(at(ws, b.rstd1)), s));
        RC(lin_fwd(dt, at(ws, b.x1), 0, E, WT(lp.fc1_w), P + lp.fc1_b, M, F, E, c.activation, nullptr, 0, at(ws, b.h), F,
                   c.activation == MIVIT_ACT_GELU ? at(ws, b.u) : nullptr, 0, s));
        RC(lin_res_ln(dt, at(ws, b.h), F, WT(lp.fc2_w), P + lp.fc2_b, M, E, F, at(ws, b.x1), at(ws, b.z2), P + lp.n2_w, P + lp.n2_b,
                      at(ws, b.x2), static_cast<float *>(at(ws, b.mean2)), static_cast<float *>(at(ws, b.rstd2)), s));
        xin = at(ws, b.x2);
    }
    // 5. final LayerNorm + readout (models.py:141, :351-354).  Only the regression-token row is normalised when it
    //    is the readout: the other rows of the final norm never reach the head.
    {
        LayerNormFwdArgs a = {};
        a.dtype = dt; a.z = xin; a.ldz = E; a.gamma = P + plan->tn_w; a.beta = P + plan->tn_b; a.E = E; a.ldy = E;
        a.mean = static_cast<float *>(at(ws, w.meanF)); a.rstd = static_cast<float *>(at(ws, w.rstdF));
        if (c.use_regression_token) {
            a.M = B; a.y = at(ws, w.pooled); a.in_rows = 1; a.in_stride = S; a.in_off = 0;
            prof_set_tag(MIVIT_PROF_LN_FWD); RC(launch_layernorm_fwd(a, s));
        } else {
            a.M = M; a.y = at(ws, w.xF);
            prof_set_tag(MIVIT_PROF_LN_FWD); RC(launch_layernorm_fwd(a, s));
            RC(launch_mean_pool_fwd(dt, at(ws, w.xF), B, S, E, at(ws, w.pooled), s));
        }
    }
    // 6. late fusion concat (models.py:356-359) and the MLP head (models.py:268-276)
    const void *head_in = at(ws, w.pooled);
    if (c.fusion == MIVIT_FUSION_LATE) {
        const int f32 = dt == MIVIT_F32;
        RC(launch_convert(f32, at(ws, w.pooled), E, f32, at(ws, w.head_in), 2 * E, B, E, 0, s, dt));
        RC(launch_convert(f32, at(ws, w.fp_out), E, f32, col_ptr(at(ws, w.head_in), E, dt), 2 * E, B, E, 0, s, dt));
        head_in = at(ws, w.head_in);
    }
    RC(lin_fwd(dt, head_in, 0, plan->head_in, WT(plan->h0_w), P + plan->h0_b, B, c.head_hidden, plan->head_in,
               MIVIT_ACT_RELU, nullptr, 0, at(ws, w.hh), c.head_hidden, nullptr, 0, s));
    RC(lin_fwd(dt, at(ws, w.hh), 0, c.head_hidden, WT(plan->h3_w), P + plan->h3_b, B, c.output_dim, c.head_hidden,
               MIVIT_ACT_NONE, nullptr, 0, out, c.output_dim, nullptr, 1, s));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
namespace {

// backward of the feature projector given d(fp_out) (rows of `dy`, leading dim lddy)
int feature_projector_bwd(const mivit_plan *plan, const Ws &w, void *ws, const float *P, float *G, const float *features,
                          int B, const void *dy, int64_t lddy, float *dfeatures, hipStream_t s) {
    const mivit_config &c = plan->c;
    const int dt = c.dtype, E = c.embed_dim, Fg = c.global_feature_dim;
    auto WT = [&](int64_t off) -> const void * {
        return dt != MIVIT_F32 ? static_cast<const void *>(static_cast<const bf16 *>(at(ws, w.wsh)) + off)
                                : static_cast<const void *>(P + off);
    };
    RC(lin_wgrad(dt, dy, lddy, at(ws, w.fp_h), 0, E, B, E, E, G + plan->fp2_w, G + plan->fp2_b, at(ws, w.wgrad),
                 w.wgrad_bytes, s));
    RC(lin_dgrad(dt, dy, lddy, WT(plan->fp2_w), B, E, E, MIVIT_ACT_RELU, at(ws, w.fp_h), E, nullptr, 0, at(ws, w.d_fp_h),
                 E, 0, s));
    RC(lin_wgrad(dt, at(ws, w.d_fp_h), E, features, 1, Fg, B, E, Fg, G + plan->fp0_w, G + plan->fp0_b, at(ws, w.wgrad),
                 w.wgrad_bytes, s));
    if (dfeatures)
        RC(lin_dgrad(dt, at(ws, w.d_fp_h), E, WT(plan->fp0_w), B, E, Fg, MIVIT_ACT_NONE, nullptr, 0, nullptr, 0, dfeatures,
                     Fg, 1, s));
    return 0;
}

}  // namespace

static int backward_impl(const mivit_plan *plan, const float *params, const float *x, const float *features, int B,
                              int T, void *workspace, size_t workspace_bytes, const float *dout, float *grads,
                              float *dfeatures, float *dx_tokens, int stage_begin, int stage_end, void *stream) {
    RC(check_call(plan, B, T, workspace_bytes, true, "mivit_backward"));
    MIVIT_CHECK(params && x && workspace && dout && grads, "mivit_backward: null pointer");
    const mivit_config &c = plan->c;
    const int nst = (int)plan->stages.size();
    MIVIT_CHECK(stage_begin >= 0 && stage_begin <= stage_end && stage_end <= nst, "mivit_backward: bad stage range [%d,%d)",
                stage_begin, stage_end);
    MIVIT_CHECK(c.fusion == MIVIT_FUSION_NONE || features, "mivit_backward: features required");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int dt = c.dtype, E = c.embed_dim, F = c.hidden_dim, H = c.num_heads, Dh = E / H, L = c.num_layers;
    const int off = c.use_regression_token ? 1 : 0, S = T + off, M = B * S, Mt = B * T;
    const int f32 = dt == MIVIT_F32;
    const Ws w = make_ws(plan, B, T, true);
    void *ws = workspace;
    const float *P = params;
    float *G = grads;
    void *wg = at(ws, w.wgrad);
    const size_t wgb = w.wgrad_bytes;
    auto WT = [&](int64_t off) -> const void * {
        return dt != MIVIT_F32 ? static_cast<const void *>(static_cast<const bf16 *>(at(ws, w.wsh)) + off)
                                : static_cast<const void *>(P + off);
    };

    slab_defer_cancel();          // (a queue left behind by a call that failed mid-layer)
    for (int st = stage_begin; st < stage_end; ++st) {
        if (st == 0) {
            // ---- head + final norm ----
            const int Hh = c.head_hidden, Hin = plan->head_in, O = c.output_dim;
            const bool fusedL = L > 0 && fused_ok(dt, E, F, H, S);
            const void *xL = L > 0 ? (fusedL ? at(ws, w.xL) : at(ws, w.layer[L - 1].x2)) : at(ws, w.x0);
            const void *head_in = c.fusion == MIVIT_FUSION_LATE ? at(ws, w.head_in) : at(ws, w.pooled);
            const void *dy = dout;
            if (!f32) { RC(launch_convert(1, dout, O, 0, at(ws, w.dout_t), O, B, O, 0, s, dt)); dy = at(ws, w.dout_t); }
            RC(lin_wgrad(dt, dy, O, at(ws, w.hh), 0, Hh, B, O, Hh, G + plan->h3_w, G + plan->h3_b, wg, wgb, s));
            RC(lin_dgrad(dt, dy, O, WT(plan->h3_w), B, O, Hh, MIVIT_ACT_RELU, at(ws, w.hh), Hh, nullptr, 0, at(ws, w.d_hh),
                         Hh, 0, s));
            RC(lin_wgrad(dt, at(ws, w.d_hh), Hh, head_in, 0, Hin, B, Hh, Hin, G + plan->h0_w, G + plan->h0_b, wg, wgb, s));
            RC(lin_dgrad(dt, at(ws, w.d_hh), Hh, WT(plan->h0_w), B, Hh, Hin, MIVIT_ACT_NONE, nullptr, 0, nullptr, 0,
                         at(ws, w.d_head_in), Hin, 0, s));
            if (c.fusion == MIVIT_FUSION_LATE)
                RC(feature_projector_bwd(plan, w, ws, P, G, features, B, col_ptr(at(ws, w.d_head_in), E, dt), Hin,
                                         dfeatures, s));
            LayerNormBwdArgs a = {};
            a.dtype = dt; a.z = xL; a.ldz = E; a.gamma = P + plan->tn_w;
            a.mean = static_cast<const float *>(at(ws, w.meanF)); a.rstd = static_cast<const float *>(at(ws, w.rstdF));
            a.E = E; a.dz = at(ws, w.dxa); a.lddz = E; a.dgamma = G + plan->tn_w; a.dbeta = G + plan->tn_b;
            a.ws = at(ws, w.ln); a.ws_bytes = w.ln_bytes;
            if (c.use_regression_token) {
                RC(launch_fill_zero(at(ws, w.dxa), (size_t)M * E * dtype_size(dt), s));
                a.dy = at(ws, w.d_head_in); a.lddy = Hin; a.M = B; a.z_rows = 1; a.z_stride = S; a.z_off = 0;
                prof_set_tag(MIVIT_PROF_LN_BWD); RC(launch_layernorm_bwd(a, s));
            } else {
                const void *dp = at(ws, w.d_head_in);
                if (Hin != E) {
                    RC(launch_convert(f32, at(ws, w.d_head_in), Hin, f32, at(ws, w.d_pool_c), E, B, E, 0, s, dt));
                    dp = at(ws, w.d_pool_c);
                }
                RC(launch_mean_pool_bwd(dt, dp, B, S, E, at(ws, w.dxb), s));
                a.dy = at(ws, w.dxb); a.lddy = E; a.M = M;
                prof_set_tag(MIVIT_PROF_LN_BWD); RC(launch_layernorm_bwd(a, s));
            }
        } else if (st <= L) {
            // ---- encoder layer l = L - st; dxa holds d(x2) on entry and d(x_in) on exit ----
            const int l = L - st;
            const LayerParams &lp = plan->layers[l];
            const Ws::L &b = w.layer[l];
            // fused layer blocks: z1 / z2 hold xhat (no means), the Linear inputs x1 / x_in exist only as xhat of the
            // producing norm: their weight gradients are taken against xhat and corrected by launch_affine_fixup
            const bool fz = fused_ok(dt, E, F, H, S);
            const void *xin = l > 0 ? (fz ? at(ws, w.layer[l - 1].z2) : at(ws, w.layer[l - 1].x2)) : at(ws, w.x0);
            LayerNormBwdArgs n2 = {};
            n2.dtype = dt; n2.dy = at(ws, w.dxa); n2.lddy = E; n2.z = at(ws, b.z2); n2.ldz = E; n2.gamma = P + lp.n2_w;
            n2.mean = fz ? nullptr : static_cast<const float *>(at(ws, b.mean2)); n2.rstd = static_cast<const float *>(at(ws, b.rstd2));
            n2.M = M; n2.E = E; n2.dz = at(ws, w.dxb); n2.lddz = E; n2.dgamma = G + lp.n2_w; n2.dbeta = G + lp.n2_b;
            n2.ws = at(ws, w.ln); n2.ws_bytes = w.ln_bytes;
            bool cs = false;
            void *dx1 = at(ws, w.dxa), *dz1 = at(ws, w.dxb);      // where d(x1) arrives / where LayerNorm-1's backward puts d(z1)
            // fused path: the three blocks' slab reductions run as ONE launch at the end of the layer (misc.hip: slab_defer_*);
            // each block then needs its own slab region.  MIVIT_NO_SLAB_DEFER=1: one reduction behind every block (A/B runs)
            static const bool no_defer = getenv("MIVIT_NO_SLAB_DEFER") != nullptr || getenv("MIVIT_NO_QKV_BWD") != nullptr;
            size_t ws_mlp = 0, ws_ao = 0;
            bool defer = false;
            if (fz) {
                const FusedOps *fo = fused_ops(dt, E);
                ws_mlp = fo->mlp_bwd_ws(M); ws_ao = fo->attn_out_bwd_ws(M);
                defer = !no_defer && wgb >= ws_mlp + ws_ao + fo->qkv_bwd_ws(M);
            }
            uint8_t *wg_ao = defer ? static_cast<uint8_t *>(wg) + ws_mlp : static_cast<uint8_t *>(wg);
            uint8_t *wg_qkv = defer ? wg_ao + ws_ao : static_cast<uint8_t *>(wg);
            const size_t wgb_ao = defer ? wgb - ws_mlp : wgb, wgb_qkv = defer ? wgb - ws_mlp - ws_ao : wgb;
            if (defer) slab_defer_begin();
            if (fz) {
                // feed-forward block in one launch (fused_bwd.hip): d(x2) -> d(x1), all six parameter gradients
                dx1 = at(ws, w.dxb); dz1 = at(ws, w.dF);
                prof_set_tag(MIVIT_PROF_MLP_BLOCK_BWD);
                RC(fused_ops(dt, E)->mlp_bwd(at(ws, w.dxa), at(ws, b.z2), static_cast<const float *>(at(ws, b.rstd2)), P + lp.n2_w,
                                        at(ws, b.z1), P + lp.n1_w, P + lp.n1_b, WT(lp.fc1_w), P + lp.fc1_b, WT(lp.fc2_w), M,
                                        c.activation, dx1, G + lp.fc1_w, G + lp.fc1_b, G + lp.fc2_w, G + lp.fc2_b, G + lp.n2_w,
                                        G + lp.n2_b, wg, wgb, s));
            } else {
            RC(ln_bwd_bias(n2, G + lp.fc2_b, &cs, s));                                            // dxb = d(z2), fc2.bias grad
            RC(lin_wgrad(dt, at(ws, w.dxb), E, at(ws, b.h), 0, F, M, E, F, G + lp.fc2_w, cs ? G + lp.fc2_b : nullptr, wg, wgb, s));
            RC(lin_dgrad(dt, at(ws, w.dxb), E, WT(lp.fc2_w), M, E, F, c.activation,
                         c.activation == MIVIT_ACT_GELU ? at(ws, b.u) : at(ws, b.h), F, nullptr, 0, at(ws, w.dF), F, 0, s));
            RC(lin_wgrad(dt, at(ws, w.dF), F, at(ws, b.x1), 0, E, M, F, E, G + lp.fc1_w, G + lp.fc1_b, wg, wgb, s));
            RC(lin_dgrad(dt, at(ws, w.dF), F, WT(lp.fc1_w), M, F, E, MIVIT_ACT_NONE, nullptr, 0, at(ws, w.dxb), E,
                         at(ws, w.dxa), E, 0, s));                                                // dxa = d(x1)
            }
            if (fz) {
                // LayerNorm-1 backward + out-projection weight / data gradient in one launch (fused_bwd.hip)
                prof_set_tag(MIVIT_PROF_ATTN_OUT_BWD);
                RC(fused_ops(dt, E)->attn_out_bwd(dx1, at(ws, b.z1), static_cast<const float *>(at(ws, b.rstd1)), P + lp.n1_w, at(ws, b.ctx),
                                       WT(lp.out_w), M, dz1, at(ws, w.dctx), G + lp.out_w, G + lp.out_b, G + lp.n1_w, G + lp.n1_b,
                                       wg_ao, wgb_ao, s));
            } else {
            LayerNormBwdArgs n1 = n2;
            n1.dy = dx1; n1.z = at(ws, b.z1); n1.gamma = P + lp.n1_w;
            n1.mean = fz ? nullptr : static_cast<const float *>(at(ws, b.mean1)); n1.rstd = static_cast<const float *>(at(ws, b.rstd1));
            n1.dz = dz1; n1.dgamma = G + lp.n1_w; n1.dbeta = G + lp.n1_b;
            RC(ln_bwd_bias(n1, G + lp.out_b, &cs, s));                                            // dxb = d(z1), out_proj.bias grad
            RC(lin_wgrad(dt, dz1, E, at(ws, b.ctx), 0, E, M, E, E, G + lp.out_w, cs ? G + lp.out_b : nullptr, wg, wgb, s));
            RC(lin_dgrad(dt, dz1, E, WT(lp.out_w), M, E, E, MIVIT_ACT_NONE, nullptr, 0, nullptr, 0,
                         at(ws, w.dctx), E, 0, s));
            }
            prof_set_tag(fz ? MIVIT_PROF_ATTN_CORE_BWD : MIVIT_PROF_ATTN_BWD);
            RC(launch_attention_bwd(dt, at(ws, b.qkv), at(ws, w.dctx), B, S, H, Dh, at(ws, w.dqkv), s));
            // q|k|v projection backward.  Fused path: weight, bias and data gradient (+ the residual branch's d(z1)) in ONE pass
            // over dqkv (fused_bwd.hip::qkv_bwd_kernel); MIVIT_NO_QKV_BWD=1 keeps the two launches it replaces (A/B runs)
            static const bool qkv_split = getenv("MIVIT_NO_QKV_BWD") != nullptr;
            int rc_q = 0;
            if (fz && !qkv_split) {
                prof_pin_tag(MIVIT_PROF_QKV_BWD);
                // (x_in of layers l > 0 is the normalised output of the layer below: the affine fix-up of the weight gradient,
                //  dW diag(gamma) + db (x) beta, rides on the kernel's slab writes)
                rc_q = fused_ops(dt, E)->qkv_bwd(at(ws, w.dqkv), xin, WT(lp.qkv_w), dz1, M, at(ws, w.dxa), G + lp.qkv_w, G + lp.qkv_b,
                                                 l > 0 ? P + plan->layers[l - 1].n2_w : nullptr, l > 0 ? P + plan->layers[l - 1].n2_b : nullptr,
                                                 wg_qkv, wgb_qkv, s);
                if (defer) { const int rc_f = slab_defer_flush(s); if (!rc_q) rc_q = rc_f; }
            } else {
                if (fz) prof_pin_tag(MIVIT_PROF_QKV_WGRAD);
                rc_q = lin_wgrad(dt, at(ws, w.dqkv), 3 * E, xin, 0, E, M, 3 * E, E, G + lp.qkv_w, G + lp.qkv_b, wg, wgb, s);
                if (!rc_q && fz && l > 0)
                    rc_q = launch_affine_fixup(G + lp.qkv_w, G + lp.qkv_b, P + plan->layers[l - 1].n2_w, P + plan->layers[l - 1].n2_b, 3 * E, E, s);
                if (fz) prof_pin_tag(MIVIT_PROF_QKV_DGRAD);
                if (!rc_q)
                    rc_q = lin_dgrad(dt, at(ws, w.dqkv), 3 * E, WT(lp.qkv_w), M, 3 * E, E, MIVIT_ACT_NONE, nullptr, 0, dz1,
                                     E, at(ws, w.dxa), E, 0, s);                                  // dxa = d(x_in)
            }
            prof_pin_tag(-1);
            RC(rc_q);
        } else {
            // ---- token assembly + embedding; dxa holds d(x0) [B,S,E] ----
            if (c.use_pos_encoding) {
                RC(launch_fill_zero(G + plan->pos, (size_t)MAX_TOKENS * E * sizeof(float), s));
                RC(launch_batch_colsum(dt, at(ws, w.dxa), B, S, E, 0, S, G + plan->pos, at(ws, w.colsum), w.colsum_bytes, s));
            }
            if (c.use_regression_token)
                RC(launch_batch_colsum(dt, at(ws, w.dxa), B, S, E, 0, 1, G + plan->reg, at(ws, w.colsum), w.colsum_bytes, s));
            if (c.fusion == MIVIT_FUSION_EARLY)
                RC(feature_projector_bwd(plan, w, ws, P, G, features, B, at(ws, w.dxa), (int64_t)S * E, dfeatures, s));
            LayerNormBwdArgs a = {};
            a.dtype = dt; a.dy = at(ws, w.dxa); a.lddy = E; a.z = at(ws, w.emb); a.ldz = E; a.gamma = P + plan->n0_w;
            a.mean = static_cast<const float *>(at(ws, w.mean0)); a.rstd = static_cast<const float *>(at(ws, w.rstd0));
            a.M = Mt; a.E = E; a.rows_per_seq = T; a.in_seq_stride = S; a.in_row_off = off;
            a.dz = at(ws, w.dxb); a.lddz = E; a.dgamma = G + plan->n0_w; a.dbeta = G + plan->n0_b;
            a.ws = at(ws, w.ln); a.ws_bytes = w.ln_bytes;
            if (c.embedding == MIVIT_EMBED_EXTERNAL) {
                prof_set_tag(MIVIT_PROF_LN_BWD); RC(launch_layernorm_bwd(a, s));                  // dxb = d(tokens)
                if (dx_tokens) RC(launch_convert(f32, at(ws, w.dxb), E, 1, dx_tokens, E, Mt, E, 0, s, dt));
            } else {
                bool cs = false;
                RC(ln_bwd_bias(a, G + plan->emb_b, &cs, s));                                      // dxb = d(embedding out)
                const int K = c.patch_size * c.patch_size;
                const StreamOps *so = stream_ops(dt);
                if (so && so->embed_ok(dt, Mt, K, E)) {
                    prof_set_tag(MIVIT_PROF_EMBED_WGRAD);
                    RC(so->embed_wgrad(at(ws, w.dxb), x, G + plan->emb_w, Mt, K, E, wg, wgb, s));
                    if (cs) RC(lin_wgrad(dt, at(ws, w.dxb), E, x, 1, K, Mt, E, K, nullptr, G + plan->emb_b, wg, wgb, s));
                } else {
                    RC(lin_wgrad(dt, at(ws, w.dxb), E, x, 1, K, Mt, E, K, G + plan->emb_w, cs ? G + plan->emb_b : nullptr, wg, wgb, s));
                }
            }
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C-ABI entry points.  Small problems are launch-bound (~100 short kernels per call): a call whose arguments were seen
// before is captured once into a hipGraph on an internal stream and replayed on the caller's stream afterwards.
// ------------------------------------------------------------------------------------------------
// "Small" = launch-bound: the kernels of a narrow model are short at row counts where those of a wide one are not, so the
// limit is on activation ELEMENTS (token rows x embedding width; 65 536 rows at E = 128, 131 072 at the reference's E = 64:
// its Framerate shape at 4096 sequences per step, ~100 launches of 10-30 us each, is replayed instead of launched).
static bool graph_sized(const mivit_plan *plan, int B, int T) {
    return plan && (int64_t)B * (T + 1) * plan->c.embed_dim <= 65536LL * 128;
}

extern "C" int mivit_forward(const mivit_plan *plan, const float *params, const float *x, const float *features, int B,
                             int T, void *workspace, size_t workspace_bytes, int need_backward, float *out,
                             void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!graph_sized(plan, B, T))
        return forward_impl(plan, params, x, features, B, T, workspace, workspace_bytes, need_backward, out, s);
    const uint64_t key[] = {1, plan->uid, (uint64_t)params, (uint64_t)x, (uint64_t)features, (uint64_t)B, (uint64_t)T,
                            (uint64_t)workspace, (uint64_t)workspace_bytes, (uint64_t)need_backward, (uint64_t)out};
    return graph_run(key, (int)(sizeof(key) / sizeof(key[0])), s, [&](hipStream_t cs) {
        return forward_impl(plan, params, x, features, B, T, workspace, workspace_bytes, need_backward, out, cs);
    });
}

extern "C" int mivit_backward(const mivit_plan *plan, const float *params, const float *x, const float *features, int B,
                              int T, void *workspace, size_t workspace_bytes, const float *dout, float *grads,
                              float *dfeatures, float *dx_tokens, int stage_begin, int stage_end, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!graph_sized(plan, B, T))
        return backward_impl(plan, params, x, features, B, T, workspace, workspace_bytes, dout, grads, dfeatures, dx_tokens,
                             stage_begin, stage_end, s);
    const uint64_t key[] = {2, plan->uid, (uint64_t)params, (uint64_t)x, (uint64_t)features, (uint64_t)B, (uint64_t)T,
                            (uint64_t)workspace, (uint64_t)workspace_bytes, (uint64_t)dout, (uint64_t)grads,
                            (uint64_t)dfeatures, (uint64_t)dx_tokens, (uint64_t)stage_begin, (uint64_t)stage_end};
    return graph_run(key, (int)(sizeof(key) / sizeof(key[0])), s, [&](hipStream_t cs) {
        return backward_impl(plan, params, x, features, B, T, workspace, workspace_bytes, dout, grads, dfeatures, dx_tokens,
                             stage_begin, stage_end, cs);
    });
}
