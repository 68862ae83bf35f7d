/*
 * mivit_hip.h -- C-ABI of libmivit_hip.so: the MI355X (gfx950) implementation of the MiViT hot path.
 *
 * The reference (Biomedical-Imaging-Group/MolecularDiffusion_MiViT) has no FFI: its path is a tree of
 * torch.nn modules in helpers/models.py.  This header is the boundary a binding would target; every entry
 * point cites the reference lines whose arithmetic it replaces.  Conventions:
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless named host_*;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued, nothing syncs;
 *   - `dtype`: MIVIT_F32 (fp32 operands, v_mfma_f32_16x16x4_f32, the 1e-4 parity mode) or MIVIT_BF16
 *     (bf16 operands / stored activations, fp32 accumulate, fp32 LayerNorm+softmax statistics);
 *     master weights, biases, LayerNorm parameters and all weight gradients are ALWAYS fp32;
 *   - return value 0 = OK, non-zero = error; mivit_last_error() returns a thread-local message;
 *   - the library never allocates or frees device memory: callers pass workspaces
 *     (sizes from mivit_*_workspace_bytes / mivit_plan_workspace_bytes).
 */
#ifndef MIVIT_HIP_H
#define MIVIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIVIT_ABI_VERSION 1

enum { MIVIT_F32 = 0, MIVIT_BF16 = 1, MIVIT_F16 = 2 };
enum { MIVIT_ACT_NONE = 0, MIVIT_ACT_RELU = 1, MIVIT_ACT_LEAKY_RELU = 2, MIVIT_ACT_GELU = 3 };
enum { MIVIT_EMBED_LINEAR = 0, MIVIT_EMBED_CNN = 1, MIVIT_EMBED_EXTERNAL = 2 };
enum { MIVIT_FUSION_NONE = 0, MIVIT_FUSION_EARLY = 1, MIVIT_FUSION_LATE = 2 };

int mivit_abi_version(void);
const char *mivit_last_error(void);
/* Number of HIP devices visible to the library (0 when there is none); never throws. */
int mivit_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Operator level.  "T" below means the element type selected by `dtype` (float or bf16).
 * ---------------------------------------------------------------------------------------------- */

/* y[M,N] = act(x[M,K] @ W[N,K]^T + bias) (+ resid[M,N]).   nn.Linear + activation
 * (helpers/models.py:37-39,57 q/k/v/out projections; :73-76 FeedForward; :164 patch embedding; :268-273 MLPHead).
 * x is T unless x_is_f32 != 0 (raw fp32 frames for the patch embedding); W, bias fp32; y, resid, y_preact T.
 * y_preact (optional) receives the pre-activation (needed by GELU backward).  ldx/ldy/ldr in elements. */
int mivit_linear_fwd(int dtype, const void *x, int x_is_f32, int64_t ldx, const float *W, const float *bias,
                     int M, int N, int K, int act, const void *resid, int64_t ldr,
                     void *y, int64_t ldy, void *y_preact, void *stream);

/* dx[M,K] = (dy[M,N] @ W[N,K]) (* act'(saved)) (+ dres[M,K]).  Autograd of nn.Linear w.r.t. its input.
 * `act`/`saved`: when act != NONE the result is multiplied by act'(.) evaluated from `saved`[M,K]
 * (post-activation for relu / leaky_relu, PRE-activation for gelu) -- backward of models.py:74. */
int mivit_linear_dgrad(int dtype, const void *dy, int64_t lddy, const float *W, int M, int N, int K,
                       int act, const void *saved, int64_t lds, const void *dres, int64_t lddr,
                       void *dx, int64_t lddx, void *stream);

/* dW[N,K] (+)= dy[M,N]^T @ x[M,K];  db[N] (+)= column sums of dy.  Deterministic (slab split over M, no atomics).
 * workspace: mivit_linear_wgrad_workspace_bytes(M,N,K).  accumulate != 0 adds into dW/db. */
size_t mivit_linear_wgrad_workspace_bytes(int M, int N, int K);
int mivit_linear_wgrad(int dtype, const void *dy, int64_t lddy, const void *x, int x_is_f32, int64_t ldx,
                       int M, int N, int K, float *dW, float *db, int accumulate,
                       void *workspace, size_t workspace_bytes, void *stream);

/* bf16-mode streaming kernels of the frame embedding (LinearProjectionEmbedding / CNNEmbedding, models.py:164,191):
 * the two launches that read the fp32 frames x[M = B*T, K = P*P].  LDS-DMA ring, see csrc/embed.hip.
 *   fwd  : y[M,E] (bf16) = x @ W^T + bias, W given as its bf16 copy [E,K]
 *   wgrad: dW[E,K] (fp32, overwritten) = dy[M,E]^T (bf16) @ x
 * Return 3 when the shape is outside the kernels' constraints (E % 128, K % 128, K >= 256, M >= 128): the caller
 * then uses mivit_linear_fwd / mivit_linear_wgrad, which accept any shape. */
int mivit_embed_fwd_bf16(const float *x, const void *W_bf16, const float *bias, int M, int K, int E, void *y_bf16,
                         void *stream);
/* Forward tiling override (0 = chosen by problem size: embed_fwd_direct2<2,4,3> when M/128 * E/128 >= 512 workgroups,
 * embed_fwd_direct<1,4,3> / <1,2,3> below that; 14 / 8 / 15 force those three, 1 / 2 / 13 the LDS-DMA designs, 3..7 the
 * other direct tilings).  For A/B runs and so that parity tests reach every launcher branch.  Returns the previous value. */
int mivit_embed_set_variant(int variant);
size_t mivit_embed_wgrad_bf16_workspace_bytes(int M, int K, int E);
int mivit_embed_wgrad_bf16(const void *dy_bf16, const float *x, int M, int K, int E, float *dW, void *workspace,
                           size_t workspace_bytes, void *stream);

/* The same two launches for SMALL frames (patch sizes up to 16 x 16 pixels; the shipped configurations use 9 x 9 = 81 and
 * 13 x 13 = 169): x[M, K] fp32 with ANY K <= 256 -- rows are only 4-byte aligned --, E = 64 or 128, M >= 256, M * K * 4 < 2^32.
 * csrc/wavestream.hip (AF32) / csrc/wgrad_small.hip (XF32): the frames are read through a raw buffer at dword alignment and
 * converted in registers.  db (optional) = column sums of dy.  Return 3 outside these constraints. */
int mivit_embed_small_supported(int M, int K, int E);
int mivit_embed_small_fwd(const float *x, const void *W_bf16, const float *bias, int M, int K, int E, void *y_bf16, void *stream);
size_t mivit_embed_small_wgrad_workspace_bytes(int M, int K, int E);
int mivit_embed_small_wgrad(const void *dy_bf16, const float *x, int M, int K, int E, float *dW, float *db, void *workspace,
                            size_t workspace_bytes, void *stream);

/* bf16-mode streaming kernels of the encoder-layer projections (csrc/rowstream.hip, csrc/wgrad_dma.hip): weights are
 * given as bf16 copies, activations are bf16, M = all tokens of the batch.  Return 3 when the shape is outside the
 * kernels' constraints (then use mivit_linear_*).
 *   rowstream_fwd  : y = act(x @ W^T + bias) (+ resid); with ln_gamma != NULL (N == 128, resid given) additionally
 *                    ln_out = LayerNorm(y) and mean/rstd per row -- the post-norm sub-layer of models.py:100-106;
 *   rowstream_dgrad: dx = (dy @ W) (* act'(saved)) (+ dres), W = [N, K] as in mivit_linear_dgrad (at most one of
 *                    saved / dres);
 *   wgrad_bf16     : dW[N, K] (fp32, overwritten) = dy[M, N]^T @ x[M, K]; db[N] (optional) = column sums of dy, taken
 *                    from the same LDS tiles. */
int mivit_rowstream_fwd(const void *x, int64_t ldx, const void *W_bf16, const float *bias, int M, int N, int K, int act,
                        const void *resid, int64_t ldr, void *y, int64_t ldy, void *y_preact, const float *ln_gamma,
                        const float *ln_beta, void *ln_out, float *mean, float *rstd, void *stream);
int mivit_rowstream_dgrad(const void *dy, int64_t lddy, const void *W_bf16, int M, int N, int K, int act,
                          const void *saved, int64_t lds, const void *dres, int64_t lddr, void *dx, int64_t lddx,
                          void *stream);
/* Same contracts as mivit_rowstream_fwd / _dgrad, "wave-stream" data movement (weight slice copied to LDS once, every
 * wave streams its own 16-row tiles straight from global memory into MFMA operands, wave-private epilogue, no barriers);
 * contraction length 128 or 256.  The engine picks per launch whichever of the two measured faster. */
int mivit_wavestream_fwd(const void *x, int64_t ldx, const void *W_bf16, const float *bias, int M, int N, int K, int act,
                         const void *resid, int64_t ldr, void *y, int64_t ldy, void *y_preact, const float *ln_gamma,
                         const float *ln_beta, void *ln_out, float *mean, float *rstd, void *stream);
int mivit_wavestream_dgrad(const void *dy, int64_t lddy, const void *W_bf16, int M, int N, int K, int act,
                           const void *saved, int64_t lds, const void *dres, int64_t lddr, void *dx, int64_t lddx,
                           void *stream);

/* Wide layers (K, N of 512-class models), bf16: LDS-DMA ring GEMMs with 256 x 128 workgroup tiles.
 * fwd:   y = act(x W^T + bias) (+ resid), optional pre-activation copy;   dgrad: dx = (dy W) * act'(saved) (+ dres).
 * N (fwd) / K (dgrad) multiple of 128, contraction length multiple of 64, M >= 256; returns 3 otherwise. */
int mivit_gemm_dma_supported(int M, int N, int K, int dgrad);
/* Tile variant of the sizing sweep (BASELINE config 4's "MFMA tile + LDS sizing"; table in DESIGN.md 4a): 0 = default,
 * 19 = first-generation 8 x (32 x 128) tile, 20..27 = transposed-product tiles.  Returns the previous value. */
int mivit_gemm_dma_set_variant(int variant);
int mivit_gemm_dma_fwd(const void *x, int64_t ldx, const void *W_bf16, const float *bias, int M, int N, int K, int act,
                       const void *resid, int64_t ldr, void *y, int64_t ldy, void *y_preact, void *stream);
int mivit_gemm_dma_dgrad(const void *dy, int64_t lddy, const void *W_bf16, int M, int N, int K, int act,
                         const void *saved, int64_t lds, const void *dres, int64_t lddr, void *dx, int64_t lddx,
                         void *stream);
size_t mivit_wgrad_bf16_workspace_bytes(int M, int N, int K);
int mivit_wgrad_bf16(const void *dy, int64_t lddy, const void *x, int64_t ldx, int M, int N, int K, float *dW, float *db,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Fused forward of the two halves of the post-norm encoder layer (helpers/models.py:97-108), bf16 mode, model width
 * E = 128, feed-forward width F = 256, 4 heads of 32 (the PSFNoise 32x64x64 configuration); csrc/fused_fwd.hip.
 * (The reference's shipped width E = 64 / F = 128 / 4 heads of 16: the ..._w64 entries further down.)
 * LayerNorm outputs travel NORMALISED: n = (z - mean) * rstd (bf16) with rstd per row (fp32); a consumer applies the
 * producing LayerNorm's affine while loading, x = gamma_in * n_in + beta_in (gamma_in = beta_in = NULL: n_in is x).
 *   attn_block_fwd: z = x + out_proj(softmax(q k^T / sqrt(32)) v), q|k|v = x Wqkv^T + bqkv   (models.py:33-59,100-102)
 *                   one wavefront per sequence of S <= 64 tokens, weights resident in LDS, q/k/v, probabilities and
 *                   the context never leave registers.  ctx [B*S, E] (the out-projection's input) is written for the
 *                   weight gradient.
 *   mlp_block_fwd : z = x + fc2(act(fc1 x))                                                     (models.py:72-77,104-106)
 *                   one wavefront per 32 rows, the hidden activations never leave registers.
 * Both: n_out = LNhat(z) (bf16), rstd[rows] (fp32).  Optional outputs (NULL = skip), for the unfused backward kernels:
 * x_out = gamma_out * n_out + beta_out, z_out, mean, qkv_out [B*S, 3E], h_out / u_out [M, F] (post- / pre-activation).
 * Weights are bf16 copies in the reference's [out, in] layout; biases and LayerNorm vectors fp32.
 * mivit_fused_layer_supported tells whether a model shape can use them. */
int mivit_fused_layer_supported(int dtype, int embed_dim, int hidden_dim, int num_heads, int tokens);
int mivit_attn_block_fwd(const void *n_in, const float *gamma_in, const float *beta_in, const void *Wqkv_bf16,
                         const float *bqkv, const void *Wo_bf16, const float *bo, const float *gamma_out,
                         const float *beta_out, int B, int S, void *ctx, void *n_out, float *rstd, void *x_out,
                         void *z_out, float *mean, void *qkv_out, void *stream);
int mivit_mlp_block_fwd(const void *n_in, const float *gamma_in, const float *beta_in, const void *W1_bf16,
                        const float *b1, const void *W2_bf16, const float *b2, const float *gamma_out,
                        const float *beta_out, int M, int act, void *n_out, float *rstd, void *x_out, void *z_out,
                        float *mean, void *h_out, void *u_out, void *stream);

/* Fused backward of the feed-forward block (autograd of models.py:72-77,104-106 in the layout above), csrc/fused_bwd.hip:
 * in : dy = dL/dx2 [M,E] bf16 (x2 = gamma2 * n2 + beta2), n2 / rstd2 = LN2's normalised output and 1/std, n1 = LN1's
 *      normalised output (the block input is x1 = gamma1 * n1 + beta1), the bf16 weight copies and fc1's bias;
 * out: dx1 = dL/dx1 [M,E] bf16;  dW1 [F,E], db1 [F], dW2 [E,F], db2 [E], dgamma2 [E], dbeta2 [E] fp32 (overwritten).
 * The hidden activations are recomputed; h, dh and the pre-norm gradient never reach HBM (three row reads, one row write).
 * Deterministic (per-workgroup slabs + fixed-order reduction).  workspace: mivit_mlp_block_bwd_workspace_bytes(M). */
size_t mivit_mlp_block_bwd_workspace_bytes(int M);
/* Which kernel runs it: 8 (default) = hidden units split over eight waves, two per SIMD; 4 = the first kernel, four waves each
 * owning a SIMD's whole register file (kept for A/B runs; same results up to the fp32 summation order of the column sums).
 * Returns the previous value. */
int mivit_mlp_block_bwd_set_waves(int waves);
int mivit_mlp_block_bwd(const void *dy, const void *n2, const float *rstd2, const float *gamma2, const void *n1,
                        const float *gamma1, const float *beta1, const void *W1_bf16, const float *b1, const void *W2_bf16,
                        int M, int act, void *dx1, float *dW1, float *db1, float *dW2, float *db2, float *dgamma2,
                        float *dbeta2, void *workspace, size_t workspace_bytes, void *stream);

/* Noise-free rendering of single-particle image sequences from trajectories (the synthetic-data step right before the hot
 * path: helpers/helpersGeneration.py:128-319 trajectories_to_video / trajectory_to_video / gaussian_2d + block_reduce, and the
 * per-PSF variant Experiments/PSFNoise/trainSettingsPSFNoise.py:196-309).  traj_px [N, T, 2] fp32 positions (x, y) in camera
 * pixels; every frame integrates npos consecutive sub-positions (optionally centred on their mean), each a Gaussian of
 * sigma sigmas[i] (fine-grid units, grid `up` times finer than the camera) whose PEAK on the fine grid is rescaled to
 * amp[n, f, p]; the fine frame is mean-pooled up x up.  out [N, nsig, T / npos, P, P] fp32.  Background, Poisson gain and
 * normalisation are element-wise torch ops on the caller's side (helpers/generation.py). */
int mivit_render_frames(const float *traj_px, int N, int T, int npos, const float *sigmas, int nsig, int P, int up,
                        const float *amp, int center, float *out, void *stream);

/* LayerNorm-1 backward + out-projection backward in one pass (autograd of x1 = LN1(x + out_proj(ctx)), models.py:57,100-102,
 * between the feed-forward block's input gradient and the attention core), csrc/fused_bwd.hip:
 * in : dy = dL/dx1 [M,E] bf16, n1 / rstd1 (LN1's normalised output, 1/std), gamma1, ctx [M,E] (out_proj's input), Wo bf16 [E,E];
 * out: dz1 = dL/d(x + out_proj(ctx)) [M,E] bf16 (also the residual branch's gradient), dctx = dz1 Wo [M,E] bf16,
 *      dWo [E,E], dbo [E], dgamma1 [E], dbeta1 [E] fp32 (overwritten).  Deterministic. */
size_t mivit_attn_out_bwd_workspace_bytes(int M);
int mivit_attn_out_bwd(const void *dy, const void *n1, const float *rstd1, const float *gamma1, const void *ctx,
                       const void *Wo_bf16, int M, void *dz1, void *dctx, float *dWo, float *dbo, float *dgamma1,
                       float *dbeta1, void *workspace, size_t workspace_bytes, void *stream);

/* q|k|v projection backward in one pass over dqkv (autograd of qkv = x Wqkv^T + bqkv, helpers/models.py:42-44, joined by the
 * residual branch's gradient, :100-102), csrc/fused_bwd.hip:
 * in : dqkv [M,3E] bf16 (from mivit_attention_bwd), x [M,E] bf16 (the projection's input rows), Wqkv bf16 [3E,E], res [M,E] bf16;
 * out: dx = dqkv Wqkv + res [M,E] bf16;  dW = dqkv^T x [3E,E], db = column sums of dqkv [3E], fp32 (overwritten).  Deterministic.
 * E = 128 (and E = 64 as ..._w64). */
size_t mivit_qkv_bwd_workspace_bytes(int M);
int mivit_qkv_bwd(const void *dqkv, const void *x, const void *Wqkv_bf16, const void *res, int M, void *dx, float *dW, float *db,
                  void *workspace, size_t workspace_bytes, void *stream);
size_t mivit_qkv_bwd_workspace_bytes_w64(int M);
int mivit_qkv_bwd_w64(const void *dqkv, const void *x, const void *Wqkv_bf16, const void *res, int M, void *dx, float *dW, float *db,
                      void *workspace, size_t workspace_bytes, void *stream);

/* The same five operators for the reference's shipped layer width: E = 64, F = 128, 4 heads of 16
 * (Experiments/Framerate/trainSettingsFramerate.py:42-47, Experiments/ImagesFeatures/...:112-188): csrc/fused_fwd.hip and
 * csrc/fused_bwd.hip compiled a second time with -DMIVIT_WIDTH64.  Same arguments, layouts and optional outputs. */
int mivit_fused_layer_supported_w64(int dtype, int embed_dim, int hidden_dim, int num_heads, int tokens);
int mivit_attn_block_fwd_w64(const void *n_in, const float *gamma_in, const float *beta_in, const void *Wqkv_bf16,
                             const float *bqkv, const void *Wo_bf16, const float *bo, const float *gamma_out,
                             const float *beta_out, int B, int S, void *ctx, void *n_out, float *rstd, void *x_out,
                             void *z_out, float *mean, void *qkv_out, void *stream);
int mivit_mlp_block_fwd_w64(const void *n_in, const float *gamma_in, const float *beta_in, const void *W1_bf16,
                            const float *b1, const void *W2_bf16, const float *b2, const float *gamma_out,
                            const float *beta_out, int M, int act, void *n_out, float *rstd, void *x_out, void *z_out,
                            float *mean, void *h_out, void *u_out, void *stream);
size_t mivit_mlp_block_bwd_workspace_bytes_w64(int M);
int mivit_mlp_block_bwd_w64(const void *dy, const void *n2, const float *rstd2, const float *gamma2, const void *n1,
                            const float *gamma1, const float *beta1, const void *W1_bf16, const float *b1, const void *W2_bf16,
                            int M, int act, void *dx1, float *dW1, float *db1, float *dW2, float *db2, float *dgamma2,
                            float *dbeta2, void *workspace, size_t workspace_bytes, void *stream);
size_t mivit_attn_out_bwd_workspace_bytes_w64(int M);
int mivit_attn_out_bwd_w64(const void *dy, const void *n1, const float *rstd1, const float *gamma1, const void *ctx,
                           const void *Wo_bf16, int M, void *dz1, void *dctx, float *dWo, float *dbo, float *dgamma1,
                           float *dbeta1, void *workspace, size_t workspace_bytes, void *stream);

/* DeepResNetEmbedding in inference mode (helpers/models.py:230-257; ResidualBlock :202-228): conv3x3(1->32)+BN+ReLU,
 * ResidualBlock(32->64), ResidualBlock(64->128), global average pool, Linear(128->E), fused in one kernel that keeps F
 * whole frames in LDS.  Eval-mode BatchNorm is folded by the caller: conv weights are pre-scaled by
 * gamma/sqrt(running_var+eps) and laid out [c_out][tap][c_in] (tap = 3*ky+kx) in the compute dtype (fp32 or bf16);
 * the biases are the folded shifts (b12/b22 = main-path shift + skip-path shift), fp32.  w0 [32][9], wfc [E][128] fp32.
 * x [N,P,P] fp32 frames, tokens [N,E] fp32.  Returns 3 if the frame side does not fit
 * (mivit_deepresnet_eval_supported tells beforehand). */
int mivit_deepresnet_eval_supported(int dtype, int patch_size);
int mivit_deepresnet_eval_fwd(int dtype, const float *x, int N, int P, int E, const float *w0, const float *b0,
                              const void *w11, const void *w12, const void *w1s, const void *w21, const void *w22,
                              const void *w2s, const float *b11, const float *b12, const float *b21, const float *b22,
                              const float *wfc, const float *bfc, float *tokens, void *stream);

/* DeepResNetEmbedding in TRAINING mode (batch-statistics BatchNorm2d, helpers/models.py:202-257), forward + backward.
 * Parameters are passed in the reference's own layouts (Conv2d weight [c_out, c_in, kh, kw] fp32, BatchNorm vectors),
 * in the order: 0 initial_conv/bn1, 1 res_block1.conv1/bn1, 2 res_block1.conv2/bn2, 3 res_block1.skip.0/.1,
 * 4 res_block2.conv1/bn1, 5 res_block2.conv2/bn2, 6 res_block2.skip.0/.1.
 * train_fwd: x [N,P,P] fp32 frames -> tokens [N,E] fp32; updates running_mean / running_var in place
 * (running = (1-momentum)*running + momentum*batch, unbiased variance; pass null to skip) and leaves the raw
 * convolution outputs + statistics in `workspace`, which train_bwd of the same step reads (same dtype, N, P, E).
 * train_bwd: dtokens [N,E] fp32 -> every parameter gradient (fp32, reference layouts, overwritten).  There is no
 * gradient w.r.t. x (frames are data).  Activations are stored in the compute dtype (fp32 or bf16), statistics,
 * pooling and the final Linear in fp32.  Returns 3 when the frame side does not fit (ask *_supported first). */
typedef struct { const float *weight, *gamma, *beta; float *running_mean, *running_var; } mivit_conv_bn;
typedef struct { mivit_conv_bn conv[7]; const float *fc_weight, *fc_bias; } mivit_deepresnet_params;
typedef struct { float *weight, *gamma, *beta; } mivit_conv_bn_grad;
typedef struct { mivit_conv_bn_grad conv[7]; float *fc_weight, *fc_bias; } mivit_deepresnet_grads;
int mivit_deepresnet_train_supported(int dtype, int patch_size);
size_t mivit_deepresnet_train_workspace_bytes(int dtype, int N, int P, int E);
int mivit_deepresnet_train_fwd(int dtype, const mivit_deepresnet_params *params, const float *x, int N, int P, int E,
                               float momentum, float eps, float *tokens, void *workspace, size_t workspace_bytes,
                               void *stream);
/* Inference with the layer-by-layer kernels (any frame side): BatchNorm uses running_mean / running_var (required,
 * not modified).  Workspace as for train_fwd. */
int mivit_deepresnet_infer(int dtype, const mivit_deepresnet_params *params, const float *x, int N, int P, int E,
                           float eps, float *tokens, void *workspace, size_t workspace_bytes, void *stream);
int mivit_deepresnet_train_bwd(int dtype, const mivit_deepresnet_params *params, const float *x, const float *dtokens,
                               int N, int P, int E, float eps, const mivit_deepresnet_grads *grads, void *workspace,
                               size_t workspace_bytes, void *stream);

/* Introspection (tests, diagnostics): byte offsets of the training workspace regions -- [0..6] raw convolution outputs
 * y0..y6 ([N*P*P, c_out] in the compute dtype), [7] forward BatchNorm tables (7 x [mean|rstd|scale|shift] x 128 fp32),
 * [8] gradient tables (7 x [k|c0|c1] x 128), [9] pooled [N,128] fp32, [10] dpooled, [11] partial sums, [12..14] the three
 * gradient buffers, [15] total bytes. */
int mivit_deepresnet_train_workspace_layout(int dtype, int N, int P, int E, size_t *offsets /* [16] */);

/* Synchronised BatchNorm for data-parallel training of the DeepResNet embedding (the reference trains on
 * one device, so its BatchNorm2d at helpers/models.py:206-225,233 always sees the whole minibatch -- this keeps that
 * true across ranks).  train_fwd / train_bwd cut into MIVIT_DEEPRESNET_STAGES stages each; call stage 0..5 in order.
 * Every stage leaves this rank's BatchNorm sums in `stats` ([2][3][128] fp64 on the device); between two stages the
 * caller all-reduces (sum) the whole buffer in the forward pass and only its first [3][128] in the backward pass (the
 * second half keeps the local sums that d gamma / d beta are made from, exactly as torch.nn.SyncBatchNorm does), then
 * calls the next stage; *global_count (fp64 on the device, read by the kernels: no host synchronisation) = (frames over
 * all ranks) * P * P, i.e. the all-reduced local N*P*P.  Not replayed as hipGraphs. */
#define MIVIT_DEEPRESNET_STAGES 6
#define MIVIT_DEEPRESNET_STATS_DOUBLES (2 * 3 * 128)
int mivit_deepresnet_train_fwd_stage(int dtype, const mivit_deepresnet_params *params, const float *x, int N, int P, int E,
                                     float momentum, float eps, float *tokens, void *workspace, size_t workspace_bytes,
                                     int stage, const double *global_count, double *stats, void *stream);
int mivit_deepresnet_train_bwd_stage(int dtype, const mivit_deepresnet_params *params, const float *x, const float *dtokens,
                                     int N, int P, int E, float eps, const mivit_deepresnet_grads *grads, void *workspace,
                                     size_t workspace_bytes, int stage, const double *global_count, double *stats, void *stream);

/* Weight gradient of narrow layers (64-wide models), bf16: dW [N,K] = dy^T x for (N, K) in {(64,64), (128,64),
 * (192,64), (64,128)}, M >= 256; the whole gradient block lives in each wave's accumulators, deterministic reduction. */
size_t mivit_wgrad_small_workspace_bytes(int M, int N, int K);
int mivit_wgrad_small(const void *dy, int64_t lddy, const void *x, int64_t ldx, int M, int N, int K, float *dW,
                      float *db /* optional: column sums of dy */, void *workspace, size_t workspace_bytes, void *stream);

/* Row LayerNorm over E, eps 1e-5, biased variance, affine (nn.LayerNorm: models.py:88-89,134,301).
 * Row r of the output goes to row  (r / rows_per_seq) * out_seq_stride + r % rows_per_seq + out_row_off  when
 * rows_per_seq > 0 (token assembly behind the regression token, models.py:347), else to row r.
 * `pos` (optional, fp32 [>=out rows per sequence, E]) is added after the affine (models.py:137-138).
 * mean/rstd (fp32 [M]) are saved for backward. */
int mivit_layernorm_fwd(int dtype, const void *z, int64_t ldz, const float *gamma, const float *beta,
                        int M, int E, void *y, int64_t ldy, int rows_per_seq, int out_seq_stride, int out_row_off,
                        const float *pos, float *mean, float *rstd, void *stream);

/* dz = LayerNorm backward; dgamma/dbeta (+)= reductions over rows.  dy rows are read through the same row map
 * as the forward wrote them.  workspace: mivit_layernorm_bwd_workspace_bytes(M,E). */
size_t mivit_layernorm_bwd_workspace_bytes(int M, int E);
int mivit_layernorm_bwd(int dtype, const void *dy, int64_t lddy, const void *z, int64_t ldz,
                        const float *gamma, const float *mean, const float *rstd, int M, int E,
                        int rows_per_seq, int in_seq_stride, int in_row_off,
                        void *dz, int64_t lddz, float *dgamma, float *dbeta, int accumulate,
                        void *workspace, size_t workspace_bytes, void *stream);

/* Multi-head self-attention core (models.py:42-54): per (b,h) softmax(q k^T / sqrt(Dh)) v, heads merged.
 * qkv: T [B*S, 3E] rows = [q | k | v] of one token (head h owns columns h*Dh..); ctx: T [B*S, E].
 * Whole sequences live in LDS (S <= mivit_attention_max_seq(dtype, Dh)). */
int mivit_attention_max_seq(int dtype, int Dh);
int mivit_attention_fwd(int dtype, const void *qkv, int B, int S, int H, int Dh, void *ctx, void *stream);
/* dqkv: T [B*S, 3E] from dctx T [B*S,E]; probabilities are recomputed from q,k (nothing saved). */
int mivit_attention_bwd(int dtype, const void *qkv, const void *dctx, int B, int S, int H, int Dh,
                        void *dqkv, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Model level: GeneralTransformer.forward / its autograd (models.py:278-361, :111-141, :81-108).
 * Parameters live in ONE fp32 arena whose layout the plan defines (so q/k/v weights are contiguous and the
 * gradient arena can be all-reduced per stage without copies).  Names are the reference state-dict keys.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mivit_config {
    int abi_version;       /* = MIVIT_ABI_VERSION */
    int dtype;             /* MIVIT_F32 | MIVIT_BF16 */
    int embedding;         /* MIVIT_EMBED_LINEAR | _CNN (same arithmetic, weight viewed [E,P*P]) | _EXTERNAL
                              (caller supplies pre-LayerNorm tokens [B,T,E], e.g. DeepResNetEmbedding) */
    int patch_size;        /* P: frame side; a token is a whole frame */
    int embed_dim;         /* E */
    int num_heads;         /* H */
    int hidden_dim;        /* F */
    int num_layers;        /* L */
    int activation;        /* MIVIT_ACT_* for FeedForward */
    int use_pos_encoding;  /* learned [1,128,E] table */
    int use_regression_token;
    int fusion;            /* MIVIT_FUSION_* */
    int global_feature_dim;
    int head_hidden;       /* MLPHead hidden (128) */
    int output_dim;        /* MLPHead outputs (1) */
} mivit_config;

typedef struct mivit_plan mivit_plan;

mivit_plan *mivit_plan_create(const mivit_config *cfg);   /* NULL on error */
void mivit_plan_destroy(mivit_plan *plan);

/* Parameter arena layout (floats). */
int mivit_plan_num_params(const mivit_plan *plan);
const char *mivit_plan_param_name(const mivit_plan *plan, int i);   /* reference state-dict key */
int64_t mivit_plan_param_offset(const mivit_plan *plan, int i);     /* float offset into the arena */
int64_t mivit_plan_param_numel(const mivit_plan *plan, int i);
int64_t mivit_plan_arena_numel(const mivit_plan *plan);
/* Backward stages, in execution order: 0 = head(+final norm, feature projector), 1..L = encoder layers
 * L-1..0, L+1 = token assembly + embedding.  Stage s owns arena floats [begin,end): its gradients are final
 * once mivit_backward(.., s, s+1, ..) has been enqueued -- this is what the data-parallel host overlaps
 * its per-stage all-reduce with. */
int mivit_plan_num_stages(const mivit_plan *plan);
int mivit_plan_stage_range(const mivit_plan *plan, int stage, int64_t *begin, int64_t *end);

size_t mivit_plan_workspace_bytes(const mivit_plan *plan, int B, int T, int need_backward);

/* x: fp32 [B,T,P,P] frames (or fp32 [B,T,E] pre-norm tokens for _EXTERNAL); features: fp32 [B,Fg] or NULL;
 * out: fp32 [B,output_dim].  The workspace keeps what backward needs; it must stay untouched until then. */
int mivit_forward(const mivit_plan *plan, const float *params, const float *x, const float *features,
                  int B, int T, void *workspace, size_t workspace_bytes, int need_backward,
                  float *out, void *stream);

/* Runs backward stages [stage_begin, stage_end).  dout: fp32 [B,output_dim].  grads: fp32 arena, same
 * layout as params; every float of a stage's range is OVERWRITTEN (no accumulation).  dfeatures (fp32
 * [B,Fg]) and dx_tokens (fp32 [B,T,E], _EXTERNAL only) may be NULL when not needed. */
int mivit_backward(const mivit_plan *plan, const float *params, const float *x, const float *features,
                   int B, int T, void *workspace, size_t workspace_bytes, const float *dout,
                   float *grads, float *dfeatures, float *dx_tokens,
                   int stage_begin, int stage_end, void *stream);

/* hipGraph replay statistics.  mivit_forward / mivit_backward / mivit_deepresnet_train_* on small (launch-bound) problems
 * capture their kernel sequence into a hipGraph the second time they see the same arguments and replay it afterwards
 * (MIVIT_GRAPHS=0 disables; off while mivit_profile_enable is active). */
void mivit_graph_stats(uint64_t *replays, uint64_t *captures, int *failures);

/* ------------------------------------------------------------------------------------------------
 * In-library kernel timing (used by bench.py for the roofline line): when a tag's bit is set in `tag_mask`, the
 * MAIN kernel of every launch in that category is bracketed by a hipEvent pair on the launch stream.
 * mivit_profile_collect waits for the recorded events, returns their summed duration and count, and resets
 * the tag.  Tags refer to the model-level engine's launches (operator-level calls are tagged MIVIT_PROF_OP).
 * ---------------------------------------------------------------------------------------------- */
enum {
    MIVIT_PROF_EMBED_FWD = 0,   /* patch-embedding GEMM  [B*T,P*P] x [E,P*P]^T            (HBM-bound)  */
    MIVIT_PROF_EMBED_WGRAD = 1, /* its weight gradient   d_emb^T x, re-reads the frames  (HBM-bound)  */
    MIVIT_PROF_LINEAR_FWD = 2,  /* qkv / out-proj / fc1 / fc2 / head forward GEMMs                     */
    MIVIT_PROF_LINEAR_DGRAD = 3,
    MIVIT_PROF_LINEAR_WGRAD = 4,
    MIVIT_PROF_ATTN_FWD = 5,
    MIVIT_PROF_ATTN_BWD = 6,
    MIVIT_PROF_LN_FWD = 7,
    MIVIT_PROF_LN_BWD = 8,
    MIVIT_PROF_OP = 9,
    /* the launches of the fused encoder-layer path (bf16, E 128 / F 256 / 4 heads), one tag per kernel: */
    MIVIT_PROF_ATTN_BLOCK_FWD = 10, /* LayerNorm affine + q|k|v + attention + out-proj + residual + LayerNorm-1 (fused_fwd.hip) */
    MIVIT_PROF_MLP_BLOCK_FWD = 11,  /* fc1 + activation + fc2 + residual + LayerNorm-2                                    */
    MIVIT_PROF_MLP_BLOCK_BWD = 12,  /* their backward incl. all six parameter gradients (fused_bwd.hip)                   */
    MIVIT_PROF_ATTN_OUT_BWD = 13,   /* LayerNorm-1 backward + out-projection data / weight gradient                       */
    MIVIT_PROF_ATTN_CORE_BWD = 14,  /* attention core backward (attention_fast.hip)                                       */
    MIVIT_PROF_QKV_WGRAD = 15,      /* q|k|v weight gradient (wgrad_dma.hip) + affine fix-up                              */
    MIVIT_PROF_QKV_DGRAD = 16,      /* q|k|v data gradient + residual gradient (rowstream.hip)                            */
    MIVIT_PROF_QKV_BWD = 17,        /* both of them in one pass over dqkv (fused_bwd.hip::qkv_bwd_kernel) + affine fix-up  */
    MIVIT_PROF_NUM_TAGS = 18
};
int mivit_profile_enable(uint64_t tag_mask);   /* 0 disables */
int mivit_profile_collect(int tag, double *total_ms, int *count);
const char *mivit_profile_tag_name(int tag);

#ifdef __cplusplus
}
#endif
#endif /* MIVIT_HIP_H */
