"""Operator-level autograd wrappers over the C-ABI kernels (include/mivit_hip.h, "Operator level").

Each function mirrors one torch op of the reference path (nn.Linear(+activation), nn.LayerNorm, the attention
core of MultiHeadAttention) and differentiates through the matching hand-written backward kernels.  Tensors must
be on the GPU; x may be float32 (fp32 MFMA), bfloat16 (bf16 MFMA, the fast path) or float16 (fp16 MFMA, general
kernels); weights / LayerNorm parameters are fp32.
"""
from __future__ import annotations

import ctypes

import torch

from . import _native as N


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return N.F32
    if t.dtype == torch.bfloat16:
        return N.BF16
    if t.dtype == torch.float16:
        return N.F16
    raise TypeError(f"unsupported dtype {t.dtype} (float32, bfloat16 or float16)")


def _gpu(*ts):
    for t in ts:
        if t is not None and t.device.type != "cuda":
            raise RuntimeError("MiViT HIP operators need GPU tensors (no CPU fallback exists in this package)")


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _s(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, act):
        _gpu(x, W, b)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        W = W.contiguous().float()
        bb = b.contiguous().float() if b is not None else None
        M, K = x2.shape
        Nn = W.shape[0]
        y = torch.empty(M, Nn, dtype=x2.dtype, device=x2.device)
        pre = torch.empty_like(y) if act == N.ACT_GELU else None
        N.check(N.lib.mivit_linear_fwd(_dt(x2), _p(x2), 0, K, _p(W), _p(bb), M, Nn, K, act, None, 0, _p(y), Nn,
                                       _p(pre), _s(x2)), "mivit_linear_fwd")
        ctx.save_for_backward(x2, W, pre if pre is not None else y)
        ctx.act, ctx.shp, ctx.has_b = act, shp, b is not None
        return y.reshape(*shp[:-1], Nn)

    @staticmethod
    def backward(ctx, dy):
        x2, W, saved = ctx.saved_tensors
        M, K = x2.shape
        Nn = W.shape[0]
        dy2 = dy.reshape(M, Nn).contiguous().to(x2.dtype)
        dt = _dt(x2)
        if ctx.act != N.ACT_NONE:   # fold act' into dy first: d(pre) = dy * act'(saved)
            # the dgrad kernel applies act' to ITS output; here act sits on the linear's output, so use a tiny
            # identity-dgrad trick-free path: elementwise in torch (glue) -- kept out of the fused engine path.
            if ctx.act == N.ACT_RELU:
                dy2 = dy2 * (saved > 0).to(dy2.dtype)
            elif ctx.act == N.ACT_LEAKY_RELU:
                dy2 = dy2 * torch.where(saved > 0, 1.0, 0.01).to(dy2.dtype)
            else:
                u = saved.float()
                cdf = 0.5 * (1 + torch.erf(u * 0.7071067811865476))
                pdf = torch.exp(-0.5 * u * u) * 0.3989422804014327
                dy2 = (dy2.float() * (cdf + u * pdf)).to(dy2.dtype)
            dy2 = dy2.contiguous()
        dx = torch.empty_like(x2)
        N.check(N.lib.mivit_linear_dgrad(dt, _p(dy2), Nn, _p(W), M, Nn, K, N.ACT_NONE, None, 0, None, 0, _p(dx), K,
                                         _s(x2)), "mivit_linear_dgrad")
        dW = torch.empty_like(W)
        db = torch.empty(Nn, dtype=torch.float32, device=W.device) if ctx.has_b else None
        wsb = N.lib.mivit_linear_wgrad_workspace_bytes(M, Nn, K)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=W.device)
        N.check(N.lib.mivit_linear_wgrad(dt, _p(dy2), Nn, _p(x2), 0, K, M, Nn, K, _p(dW), _p(db), 0, _p(ws), ws.numel(),
                                         _s(x2)), "mivit_linear_wgrad")
        return dx.reshape(ctx.shp), dW, db, None


def linear(x, W, b=None, act=N.ACT_NONE):
    """act(x @ W^T + b): nn.Linear (+ relu / leaky_relu / gelu)."""
    return _Linear.apply(x, W, b, act)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        _gpu(x, w, b)
        shp = x.shape
        E = shp[-1]
        x2 = x.reshape(-1, E).contiguous()
        w, b = w.contiguous().float(), b.contiguous().float()
        M = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(M, dtype=torch.float32, device=x2.device)
        rstd = torch.empty_like(mean)
        N.check(N.lib.mivit_layernorm_fwd(_dt(x2), _p(x2), E, _p(w), _p(b), M, E, _p(y), E, 0, 0, 0, None, _p(mean),
                                          _p(rstd), _s(x2)), "mivit_layernorm_fwd")
        ctx.save_for_backward(x2, w, mean, rstd)
        ctx.shp = shp
        return y.reshape(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, w, mean, rstd = ctx.saved_tensors
        M, E = x2.shape
        dy2 = dy.reshape(M, E).contiguous().to(x2.dtype)
        dx = torch.empty_like(x2)
        dg = torch.empty(E, dtype=torch.float32, device=x2.device)
        db = torch.empty_like(dg)
        wsb = N.lib.mivit_layernorm_bwd_workspace_bytes(M, E)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=x2.device)
        N.check(N.lib.mivit_layernorm_bwd(_dt(x2), _p(dy2), E, _p(x2), E, _p(w), _p(mean), _p(rstd), M, E, 0, 0, 0,
                                          _p(dx), E, _p(dg), _p(db), 0, _p(ws), ws.numel(), _s(x2)),
                "mivit_layernorm_bwd")
        return dx.reshape(ctx.shp), dg, db


def layer_norm(x, weight, bias):
    """nn.LayerNorm over the last dimension (eps 1e-5)."""
    return _LayerNorm.apply(x, weight, bias)


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, H):
        _gpu(qkv)
        B, S, E3 = qkv.shape
        E = E3 // 3
        qkv = qkv.contiguous()
        out = torch.empty(B, S, E, dtype=qkv.dtype, device=qkv.device)
        N.check(N.lib.mivit_attention_fwd(_dt(qkv), _p(qkv), B, S, H, E // H, _p(out), _s(qkv)), "mivit_attention_fwd")
        ctx.save_for_backward(qkv)
        ctx.H = H
        return out

    @staticmethod
    def backward(ctx, dctx):
        (qkv,) = ctx.saved_tensors
        B, S, E3 = qkv.shape
        E = E3 // 3
        dctx = dctx.contiguous().to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        N.check(N.lib.mivit_attention_bwd(_dt(qkv), _p(qkv), _p(dctx), B, S, ctx.H, E // ctx.H, _p(dqkv), _s(qkv)),
                "mivit_attention_bwd")
        return dqkv, None


def attention(qkv, num_heads):
    """softmax(q k^T / sqrt(Dh)) v per head, heads merged.  qkv: [B, S, 3E] = [q | k | v] per token."""
    return _Attention.apply(qkv, num_heads)


def deepresnet_eval_supported(dtype: torch.dtype, patch_size: int) -> bool:
    return bool(N.lib.mivit_deepresnet_eval_supported(N.BF16 if dtype == torch.bfloat16 else N.F32, int(patch_size)))


def deepresnet_eval(x: torch.Tensor, pack: dict, embed_dim: int) -> torch.Tensor:
    """Inference-mode DeepResNetEmbedding (reference models.py:230-257) in one fused kernel.
    x [N,P,P] fp32 frames; ``pack`` = BN-folded weights from ``DeepResNetEmbedding.folded``; returns [N,E] fp32."""
    _gpu(x)
    x = x.contiguous().float()
    n, p, p2 = x.shape
    assert p == p2, "square frames expected"
    out = torch.empty(n, embed_dim, device=x.device, dtype=torch.float32)
    if n == 0:
        return out
    order = ("w0", "b0", "w11", "w12", "w1s", "w21", "w22", "w2s", "b11", "b12", "b21", "b22", "wfc", "bfc")
    N.check(N.lib.mivit_deepresnet_eval_fwd(_dt(pack["w11"]), _p(x), n, p, embed_dim, *[_p(pack[k]) for k in order],
                                            _p(out), _s(x)), "mivit_deepresnet_eval_fwd")
    return out


def deepresnet_train_supported(dtype: torch.dtype, patch_size: int) -> bool:
    return bool(N.lib.mivit_deepresnet_train_supported(N.BF16 if dtype == torch.bfloat16 else N.F32, int(patch_size)))


class _DeepResNetTrain(torch.autograd.Function):
    """Training-mode DeepResNetEmbedding (reference models.py:230-257) on the hand-written conv / BatchNorm kernels.
    Inputs after ``x``: for each of the 7 conv+BN pairs (weight, gamma, beta), then fc.weight, fc.bias  (23 tensors)."""

    @staticmethod
    def forward(ctx, x, dtype_code, momentum, eps, running, *params):
        n, p, _ = x.shape
        e = params[21].shape[0]
        prm = N.DeepResNetParams()
        for i in range(7):
            w, g, b = params[3 * i:3 * i + 3]
            rm, rv = running[i]
            prm.conv[i] = N.ConvBn(w.data_ptr(), g.data_ptr(), b.data_ptr(), rm.data_ptr() if rm is not None else None,
                                   rv.data_ptr() if rv is not None else None)
        prm.fc_weight, prm.fc_bias = params[21].data_ptr(), params[22].data_ptr()
        nbytes = N.lib.mivit_deepresnet_train_workspace_bytes(dtype_code, n, p, e)
        if nbytes == 0:
            raise N.MivitError(f"DeepResNet training kernels do not support frame side {p}")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        tokens = torch.empty(n, e, device=x.device, dtype=torch.float32)
        N.check(N.lib.mivit_deepresnet_train_fwd(dtype_code, ctypes.addressof(prm), _p(x), n, p, e, momentum, eps, _p(tokens),
                                                 _p(ws), nbytes, _s(x)), "mivit_deepresnet_train_fwd")
        ctx.save_for_backward(x, ws, *params)
        ctx.meta = (dtype_code, n, p, e, eps, nbytes)
        return tokens

    @staticmethod
    def backward(ctx, dtokens):
        x, ws, *params = ctx.saved_tensors
        dtype_code, n, p, e, eps, nbytes = ctx.meta
        prm, gr = N.DeepResNetParams(), N.DeepResNetGrads()
        grads = [torch.empty_like(t) for t in params]
        for i in range(7):
            w, g, b = params[3 * i:3 * i + 3]
            prm.conv[i] = N.ConvBn(w.data_ptr(), g.data_ptr(), b.data_ptr(), None, None)
            gr.conv[i] = N.ConvBnGrad(*[t.data_ptr() for t in grads[3 * i:3 * i + 3]])
        prm.fc_weight, prm.fc_bias = params[21].data_ptr(), params[22].data_ptr()
        gr.fc_weight, gr.fc_bias = grads[21].data_ptr(), grads[22].data_ptr()
        dtokens = dtokens.contiguous().float()
        N.check(N.lib.mivit_deepresnet_train_bwd(dtype_code, ctypes.addressof(prm), _p(x), _p(dtokens), n, p, e, eps,
                                                 ctypes.addressof(gr), _p(ws), nbytes, _s(x)), "mivit_deepresnet_train_bwd")
        return (None, None, None, None, None, *grads)


DEEPRESNET_STAGES = 6


class _DeepResNetTrainSync(torch.autograd.Function):
    """_DeepResNetTrain with BatchNorm statistics synchronised over a process group: the native forward /
    backward run stage by stage with one small fp64 all-reduce of the BatchNorm sums between stages, so every rank
    normalises with the statistics of the whole minibatch -- what the single-device reference (models.py:206-225,233)
    computes.  d gamma / d beta stay per-rank sums like every other parameter gradient (the DP all-reduce averages them)."""

    @staticmethod
    def forward(ctx, x, dtype_code, momentum, eps, running, group, *params):
        import torch.distributed as dist
        n, p, _ = x.shape
        e = params[21].shape[0]
        prm = N.DeepResNetParams()
        for i in range(7):
            w, g, b = params[3 * i:3 * i + 3]
            rm, rv = running[i]
            prm.conv[i] = N.ConvBn(w.data_ptr(), g.data_ptr(), b.data_ptr(), rm.data_ptr() if rm is not None else None,
                                   rv.data_ptr() if rv is not None else None)
        prm.fc_weight, prm.fc_bias = params[21].data_ptr(), params[22].data_ptr()
        nbytes = N.lib.mivit_deepresnet_train_workspace_bytes(dtype_code, n, p, e)
        if nbytes == 0:
            raise N.MivitError(f"DeepResNet training kernels do not support frame side {p}")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        tokens = torch.empty(n, e, device=x.device, dtype=torch.float32)
        count = torch.full((1,), float(n * p * p), dtype=torch.float64, device=x.device)
        dist.all_reduce(count, group=group)
        stats = torch.zeros(2, 3, 128, dtype=torch.float64, device=x.device)
        for stage in range(DEEPRESNET_STAGES):
            N.check(N.lib.mivit_deepresnet_train_fwd_stage(dtype_code, ctypes.addressof(prm), _p(x), n, p, e, momentum, eps,
                                                           _p(tokens), _p(ws), nbytes, stage, _p(count), _p(stats), _s(x)),
                    "mivit_deepresnet_train_fwd_stage")
            if stage + 1 < DEEPRESNET_STAGES:
                dist.all_reduce(stats, group=group)
        ctx.save_for_backward(x, ws, count, *params)
        ctx.meta = (dtype_code, n, p, e, eps, nbytes, group)
        return tokens

    @staticmethod
    def backward(ctx, dtokens):
        import torch.distributed as dist
        x, ws, count, *params = ctx.saved_tensors
        dtype_code, n, p, e, eps, nbytes, group = ctx.meta
        prm, gr = N.DeepResNetParams(), N.DeepResNetGrads()
        grads = [torch.empty_like(t) for t in params]
        for i in range(7):
            w, g, b = params[3 * i:3 * i + 3]
            prm.conv[i] = N.ConvBn(w.data_ptr(), g.data_ptr(), b.data_ptr(), None, None)
            gr.conv[i] = N.ConvBnGrad(*[t.data_ptr() for t in grads[3 * i:3 * i + 3]])
        prm.fc_weight, prm.fc_bias = params[21].data_ptr(), params[22].data_ptr()
        gr.fc_weight, gr.fc_bias = grads[21].data_ptr(), grads[22].data_ptr()
        dtokens = dtokens.contiguous().float()
        stats = torch.zeros(2, 3, 128, dtype=torch.float64, device=x.device)
        for stage in range(DEEPRESNET_STAGES):
            N.check(N.lib.mivit_deepresnet_train_bwd_stage(dtype_code, ctypes.addressof(prm), _p(x), _p(dtokens), n, p, e, eps,
                                                           ctypes.addressof(gr), _p(ws), nbytes, stage, _p(count), _p(stats),
                                                           _s(x)), "mivit_deepresnet_train_bwd_stage")
            if stage + 1 < DEEPRESNET_STAGES:
                dist.all_reduce(stats[0], group=group)       # stats[1] keeps this rank's sums (d gamma / d beta)
        return (None, None, None, None, None, None, *grads)


def deepresnet_train_sync(x, dtype, momentum, eps, running, params, group=None):
    """deepresnet_train with BatchNorm statistics taken over every rank of ``group`` (default process group)."""
    _gpu(x, *params)
    for t in params:
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("DeepResNet parameters must be contiguous float32 tensors")
    return _DeepResNetTrainSync.apply(x.contiguous().float(), N.BF16 if dtype == torch.bfloat16 else N.F32, float(momentum),
                                      float(eps), running, group, *params)


def deepresnet_train(x, dtype, momentum, eps, running, params):
    """x [N,P,P] fp32 frames -> tokens [N,E] fp32; ``running`` = 7 pairs (running_mean, running_var) updated in place
    (or (None, None)); ``params`` = the 23 parameter tensors (fp32, contiguous, reference layouts)."""
    _gpu(x, *params)
    for t in params:
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("DeepResNet parameters must be contiguous float32 tensors")
    return _DeepResNetTrain.apply(x.contiguous().float(), N.BF16 if dtype == torch.bfloat16 else N.F32, float(momentum),
                                  float(eps), running, *params)


@torch.no_grad()
def deepresnet_infer(x, dtype, eps, running, params, chunk_frames: int = 8192):
    """Inference-mode DeepResNetEmbedding on the layer-by-layer kernels (any frame side; running statistics).
    Frames are processed in chunks so the activation workspace stays bounded."""
    _gpu(x, *params)
    x = x.contiguous().float()
    n, p, _ = x.shape
    e = params[21].shape[0]
    code = N.BF16 if dtype == torch.bfloat16 else N.F32
    prm = N.DeepResNetParams()
    for i in range(7):
        w, g, b = params[3 * i:3 * i + 3]
        prm.conv[i] = N.ConvBn(w.data_ptr(), g.data_ptr(), b.data_ptr(), running[i][0].data_ptr(), running[i][1].data_ptr())
    prm.fc_weight, prm.fc_bias = params[21].data_ptr(), params[22].data_ptr()
    out = torch.empty(n, e, device=x.device, dtype=torch.float32)
    ws = None
    for f0 in range(0, n, chunk_frames):
        m = min(chunk_frames, n - f0)
        nbytes = N.lib.mivit_deepresnet_train_workspace_bytes(code, m, p, e)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        N.check(N.lib.mivit_deepresnet_infer(code, ctypes.addressof(prm), _p(x[f0:f0 + m]), m, p, e, float(eps), _p(out[f0:f0 + m]),
                                             _p(ws), ws.numel(), _s(x)), "mivit_deepresnet_infer")
    return out


# ---------------------------------------------------------------------------------------------------------------
# fused encoder-layer blocks (csrc/fused_fwd.hip / fused_bwd.hip), bf16 mode, 4 heads, E = 128 / F = 256 or E = 64 / F = 128
# ---------------------------------------------------------------------------------------------------------------
def _fused_entry(name: str, embed_dim: int, hidden_dim=None):
    """the C entry point of a fused block for this layer width (the kernels are compiled per width: anything else is refused
    here, before a launch that would index past the tensors)"""
    if embed_dim not in (64, 128) or (hidden_dim is not None and hidden_dim != 2 * embed_dim):
        raise ValueError(f"{name}: fused blocks exist for E=128/F=256 and E=64/F=128, got E={embed_dim}"
                         + (f" F={hidden_dim}" if hidden_dim is not None else ""))
    return getattr(N.lib, name + ("_w64" if embed_dim == 64 else ""))


def fused_layer_supported(embed_dim: int, hidden_dim: int, num_heads: int, tokens: int) -> bool:
    if embed_dim not in (64, 128):
        return False
    return bool(_fused_entry("mivit_fused_layer_supported", embed_dim)(N.BF16, embed_dim, hidden_dim, num_heads, tokens))


def _f32(t):
    return None if t is None else t.contiguous().float()


@torch.no_grad()
def attn_block_fwd(n_in, gamma_in, beta_in, Wqkv, bqkv, Wo, bo, gamma_out, beta_out, extras=False):
    """n_in [B,S,E] bf16 (normalised tokens; x = gamma_in * n_in + beta_in, or n_in itself when gamma_in is None);
    Wqkv [3E,E] / Wo [E,E] bf16; returns dict(n, rstd, ctx[, x, z, mean, qkv]) -- reference models.py:33-59,100-102."""
    _gpu(n_in, Wqkv, Wo)
    B, S, E = n_in.shape
    n_in = n_in.contiguous()
    dev = n_in.device
    out = {"n": torch.empty(B, S, E, dtype=torch.bfloat16, device=dev), "rstd": torch.empty(B, S, device=dev),
           "ctx": torch.empty(B, S, E, dtype=torch.bfloat16, device=dev)}
    if extras:
        out.update(x=torch.empty_like(out["n"]), z=torch.empty_like(out["n"]), mean=torch.empty(B, S, device=dev),
                   qkv=torch.empty(B, S, 3 * E, dtype=torch.bfloat16, device=dev))
    gi, bi, go, bo_ = _f32(gamma_in), _f32(beta_in), _f32(gamma_out), _f32(beta_out)
    bq, bo2 = _f32(bqkv), _f32(bo)
    if tuple(Wqkv.shape) != (3 * E, E) or tuple(Wo.shape) != (E, E):
        raise ValueError(f"attn_block_fwd: weights {tuple(Wqkv.shape)}, {tuple(Wo.shape)} do not match E={E}")
    N.check(_fused_entry("mivit_attn_block_fwd", E)(_p(n_in), _p(gi), _p(bi), _p(Wqkv.contiguous()), _p(bq), _p(Wo.contiguous()), _p(bo2),
                                       _p(go), _p(bo_), B, S, _p(out["ctx"]), _p(out["n"]), _p(out["rstd"]), _p(out.get("x")),
                                       _p(out.get("z")), _p(out.get("mean")), _p(out.get("qkv")), _s(n_in)),
            "mivit_attn_block_fwd")
    return out


@torch.no_grad()
def mlp_block_fwd(n_in, gamma_in, beta_in, W1, b1, W2, b2, gamma_out, beta_out, act=N.ACT_RELU, extras=False):
    """n_in [M,E] bf16; W1 [F,E], W2 [E,F] bf16; returns dict(n, rstd[, x, z, mean, h, u]) -- models.py:72-77,104-106."""
    _gpu(n_in, W1, W2)
    M, E = n_in.shape
    Fh = W1.shape[0]
    n_in = n_in.contiguous()
    dev = n_in.device
    out = {"n": torch.empty(M, E, dtype=torch.bfloat16, device=dev), "rstd": torch.empty(M, device=dev)}
    if extras:
        out.update(x=torch.empty_like(out["n"]), z=torch.empty_like(out["n"]), mean=torch.empty(M, device=dev),
                   h=torch.empty(M, Fh, dtype=torch.bfloat16, device=dev), u=torch.empty(M, Fh, dtype=torch.bfloat16, device=dev))
    gi, bi, go, bo_ = _f32(gamma_in), _f32(beta_in), _f32(gamma_out), _f32(beta_out)
    b1f, b2f = _f32(b1), _f32(b2)
    if tuple(W1.shape) != (Fh, E) or tuple(W2.shape) != (E, Fh):
        raise ValueError(f"mlp_block_fwd: weights {tuple(W1.shape)}, {tuple(W2.shape)} do not match E={E}")
    N.check(_fused_entry("mivit_mlp_block_fwd", E, Fh)(_p(n_in), _p(gi), _p(bi), _p(W1.contiguous()), _p(b1f), _p(W2.contiguous()), _p(b2f),
                                      _p(go), _p(bo_), M, act, _p(out["n"]), _p(out["rstd"]), _p(out.get("x")), _p(out.get("z")),
                                      _p(out.get("mean")), _p(out.get("h")), _p(out.get("u")), _s(n_in)), "mivit_mlp_block_fwd")
    return out


@torch.no_grad()
def mlp_block_bwd(dy, n2, rstd2, gamma2, n1, gamma1, beta1, W1, b1, W2, act=N.ACT_RELU):
    """Backward of the feed-forward block (see include/mivit_hip.h): returns dict(dx1, dW1, db1, dW2, db2, dgamma2, dbeta2)."""
    _gpu(dy, n2, n1, W1, W2)
    M, E = dy.shape
    Fh = W1.shape[0]
    dev = dy.device
    out = {"dx1": torch.empty(M, E, dtype=torch.bfloat16, device=dev), "dW1": torch.empty(Fh, E, device=dev),
           "db1": torch.empty(Fh, device=dev), "dW2": torch.empty(E, Fh, device=dev), "db2": torch.empty(E, device=dev),
           "dgamma2": torch.empty(E, device=dev), "dbeta2": torch.empty(E, device=dev)}
    if tuple(W1.shape) != (Fh, E) or tuple(W2.shape) != (E, Fh):
        raise ValueError(f"mlp_block_bwd: weights {tuple(W1.shape)}, {tuple(W2.shape)} do not match E={E}")
    nbytes = _fused_entry("mivit_mlp_block_bwd_workspace_bytes", E, Fh)(M)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    args = [dy.contiguous(), n2.contiguous(), _f32(rstd2), _f32(gamma2), n1.contiguous(), _f32(gamma1), _f32(beta1), W1.contiguous(),
            _f32(b1), W2.contiguous()]
    N.check(_fused_entry("mivit_mlp_block_bwd", E, Fh)(*[_p(t) for t in args], M, act, _p(out["dx1"]), _p(out["dW1"]), _p(out["db1"]), _p(out["dW2"]),
                                      _p(out["db2"]), _p(out["dgamma2"]), _p(out["dbeta2"]), _p(ws), nbytes, _s(dy)),
            "mivit_mlp_block_bwd")
    return out


@torch.no_grad()
def attn_out_bwd(dy, n1, rstd1, gamma1, ctx, Wo):
    """LayerNorm-1 backward + out-projection backward (include/mivit_hip.h): dict(dz1, dctx, dWo, dbo, dgamma1, dbeta1)."""
    _gpu(dy, n1, ctx, Wo)
    M, E = dy.shape
    dev = dy.device
    out = {"dz1": torch.empty(M, E, dtype=torch.bfloat16, device=dev), "dctx": torch.empty(M, E, dtype=torch.bfloat16, device=dev),
           "dWo": torch.empty(E, E, device=dev), "dbo": torch.empty(E, device=dev), "dgamma1": torch.empty(E, device=dev),
           "dbeta1": torch.empty(E, device=dev)}
    if tuple(Wo.shape) != (E, E):
        raise ValueError(f"attn_out_bwd: weight {tuple(Wo.shape)} does not match E={E}")
    nbytes = _fused_entry("mivit_attn_out_bwd_workspace_bytes", E)(M)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    args = [dy.contiguous(), n1.contiguous(), _f32(rstd1), _f32(gamma1), ctx.contiguous(), Wo.contiguous()]
    N.check(_fused_entry("mivit_attn_out_bwd", E)(*[_p(t) for t in args], M, _p(out["dz1"]), _p(out["dctx"]), _p(out["dWo"]), _p(out["dbo"]),
                                     _p(out["dgamma1"]), _p(out["dbeta1"]), _p(ws), nbytes, _s(dy)), "mivit_attn_out_bwd")
    return out


@torch.no_grad()
def qkv_bwd(dqkv, x, Wqkv, res):
    """q|k|v projection backward in one pass over dqkv (include/mivit_hip.h): dict(dx, dW, db); dx = dqkv Wqkv + res."""
    _gpu(dqkv, x, Wqkv, res)
    M, E = x.shape
    if tuple(dqkv.shape) != (M, 3 * E) or tuple(Wqkv.shape) != (3 * E, E) or tuple(res.shape) != (M, E):
        raise ValueError(f"qkv_bwd: shapes {tuple(dqkv.shape)}, {tuple(Wqkv.shape)}, {tuple(res.shape)} do not match x {tuple(x.shape)}")
    dev = x.device
    out = {"dx": torch.empty(M, E, dtype=torch.bfloat16, device=dev), "dW": torch.empty(3 * E, E, device=dev),
           "db": torch.empty(3 * E, device=dev)}
    nbytes = _fused_entry("mivit_qkv_bwd_workspace_bytes", E)(M)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    args = [dqkv.contiguous(), x.contiguous(), Wqkv.contiguous(), res.contiguous()]
    N.check(_fused_entry("mivit_qkv_bwd", E)(*[_p(t) for t in args], M, _p(out["dx"]), _p(out["dW"]), _p(out["db"]), _p(ws), nbytes,
                                             _s(x)), "mivit_qkv_bwd")
    return out
