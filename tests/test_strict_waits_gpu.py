"""Counted waits vs blanket waits, bitwise.

`libmivit_hip_strict.so` is the same library compiled with -DMIVIT_STRICT_WAITS: every `wait_vm<N>()` (a hand-counted
`s_waitcnt vmcnt(N)`, csrc/stream_prims.h) becomes `vmcnt(0)`.  Arithmetic and summation order are identical, so the
two libraries must agree BITWISE on every output; a count that lets a tile, a weight stage or a register stage arrive
late shows as a difference.  One run at the shapes bench.py times (every persistent loop >= 2 iterations, ragged last tiles),
through the C-ABI of both libraries -- not a soak: the VM program orders themselves are checked on the host by
tests/test_wait_model.py and the generated ISA by scripts/isa_check.py.
"""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import mivit_oracle as orc
from util import build_product_model

pytestmark = pytest.mark.gpu

E, FH = 128, 256


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _randn(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, generator=g, device="cuda") * scale


def _bf(t):
    return t.to(torch.bfloat16)


def _run_everything():
    """Every kernel that orders memory with counted waits, at bench-scale shapes, through whatever library `_native.lib` is."""
    from moleculardiffusion_mivit_amd import _native as N, ops
    out = {}
    M = 140000 + 17
    # --- frame embedding: forward (direct2 by size; the small-problem tilings forced) and weight gradient ---
    Me, K = 65536 + 31, 4096
    x = _randn((Me, K), 1)
    W = _bf(_randn((E, K), 2, 1 / math.sqrt(K)))
    b = _randn((E,), 3)
    for variant, rows in ((0, Me), (8, 3000), (15, 1000), (3, 3000), (1, 3000), (13, 3000)):
        old = N.lib.mivit_embed_set_variant(variant)
        y = torch.empty(rows, E, dtype=torch.bfloat16, device="cuda")
        N.check(N.lib.mivit_embed_fwd_bf16(_p(x), _p(W), _p(b), rows, K, E, _p(y), _st()), "embed_fwd")
        N.lib.mivit_embed_set_variant(old)
        out[f"embed_fwd_v{variant}"] = y
    dy = _bf(_randn((Me, E), 4))
    ws = torch.empty(max(N.lib.mivit_embed_wgrad_bf16_workspace_bytes(Me, K, E), 16), dtype=torch.uint8, device="cuda")
    dW = torch.empty(E, K, device="cuda")
    N.check(N.lib.mivit_embed_wgrad_bf16(_p(dy), _p(x), Me, K, E, _p(dW), _p(ws), ws.numel(), _st()), "embed_wgrad")
    out["embed_wgrad"] = dW
    del x, dy
    # --- row-stream GEMMs: forward slices (+ residual + LayerNorm), data gradients incl. the K = 384 q|k|v one ---
    a128, a256, a384 = _bf(_randn((M, 128), 5)), _bf(_randn((M, 256), 6)), _bf(_randn((M, 384), 7))
    for name, (A, Nn, Kk, dgrad) in {"qkv_fwd": (a128, 384, 128, False), "fc2_ln_fwd": (a256, 128, 256, False),
                                     "qkv_dgrad": (a384, 384, 128, True), "fc1_dgrad": (a256, 256, 128, True),
                                     "fc2_dgrad": (a128, 128, 256, True)}.items():
        Wt = _bf(_randn((Nn, Kk), 8, 0.1))
        if not dgrad:
            y = torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda")
            ln = Nn == 128
            r = _bf(_randn((M, Nn), 9)) if ln else None
            lno = torch.empty_like(y) if ln else None
            mean, rstd = (torch.empty(M, device="cuda"), torch.empty(M, device="cuda")) if ln else (None, None)
            gam, bet = (torch.ones(Nn, device="cuda"), torch.zeros(Nn, device="cuda")) if ln else (None, None)
            bias = _randn((Nn,), 10)
            N.check(N.lib.mivit_rowstream_fwd(_p(A), Kk, _p(Wt), _p(bias), M, Nn, Kk, 0, _p(r), Nn, _p(y), Nn, None,
                                              _p(gam), _p(bet), _p(lno), _p(mean), _p(rstd), _st()), name)
            out[name] = y
            if ln:
                out[name + "_ln"] = lno
        else:
            dx = torch.empty(M, Kk, dtype=torch.bfloat16, device="cuda")
            res = _bf(_randn((M, Kk), 11))
            N.check(N.lib.mivit_rowstream_dgrad(_p(A), Nn, _p(Wt), M, Nn, Kk, 0, None, Kk, _p(res), Kk, _p(dx), Kk, _st()), name)
            out[name] = dx
    ws = torch.empty(max(N.lib.mivit_wgrad_bf16_workspace_bytes(M, 384, 128), 16), dtype=torch.uint8, device="cuda")
    dWq, dbq = torch.empty(384, 128, device="cuda"), torch.empty(384, device="cuda")
    N.check(N.lib.mivit_wgrad_bf16(_p(a384), 384, _p(a128), 128, M, 384, 128, _p(dWq), _p(dbq), _p(ws), ws.numel(), _st()), "wgrad")
    out["qkv_wgrad"] = dWq
    # --- wide-layer LDS-DMA GEMMs (BASELINE config 4 shapes), every tile structure ---
    Mw = 9000 + 13
    xw, Ww = _bf(_randn((Mw, 512), 12)), _bf(_randn((1536, 512), 13, 0.05))
    for v in (0, 19, 21, 23, 31, 33):
        old = N.lib.mivit_gemm_dma_set_variant(v)
        y = torch.empty(Mw, 1536, dtype=torch.bfloat16, device="cuda")
        N.check(N.lib.mivit_gemm_dma_fwd(_p(xw), 512, _p(Ww), None, Mw, 1536, 512, 0, None, 0, _p(y), 1536, None, _st()), "gemm_dma_fwd")
        dxw = torch.empty(Mw, 512, dtype=torch.bfloat16, device="cuda")
        N.check(N.lib.mivit_gemm_dma_dgrad(_p(y), 1536, _p(Ww), Mw, 1536, 512, 0, None, 0, None, 0, _p(dxw), 512, _st()), "gemm_dma_dgrad")
        N.lib.mivit_gemm_dma_set_variant(old)
        out[f"gemm_dma_fwd_v{v}"], out[f"gemm_dma_dgrad_v{v}"] = y, dxw
    # --- fused backward blocks ---
    for Ew, FHw in ((E, FH), (64, 128)):          # both layer widths the blocks are compiled for (csrc/elem.h)
        args = (_bf(_randn((M, Ew), 21)), _bf(_randn((M, Ew), 22)), 1 + 0.1 * _randn((M,), 23).abs(), 1 + 0.1 * _randn((Ew,), 24),
                _bf(_randn((M, Ew), 25)), 1 + 0.1 * _randn((Ew,), 26), 0.1 * _randn((Ew,), 27), _bf(_randn((FHw, Ew), 28, 0.1)),
                0.1 * _randn((FHw,), 29), _bf(_randn((Ew, FHw), 30, 0.1)))
        for waves in (8, 4) if Ew == 128 else (4,):          # both kernels behind the entry (hidden units split over 8 / 4 waves)
            old = N.lib.mivit_mlp_block_bwd_set_waves(waves)
            for k, v in ops.mlp_block_bwd(*args).items():
                out[f"mlp_bwd{waves}_w{Ew}_" + k] = v
            N.lib.mivit_mlp_block_bwd_set_waves(old)
        for k, v in ops.attn_out_bwd(args[0], args[1], args[2], args[3], args[4], _bf(_randn((Ew, Ew), 31, 0.1))).items():
            out[f"attn_out_bwd_w{Ew}_" + k] = v
        for k, v in ops.qkv_bwd(_bf(_randn((M, 3 * Ew), 39)), args[4], _bf(_randn((3 * Ew, Ew), 40, 0.1)), args[0]).items():
            out[f"qkv_bwd_w{Ew}_" + k] = v
        # the attention block's forward (no counted waits of its own: rides along as a repeatability check at this scale)
        for S_ in (31, 61):
            Bq = 2600
            o = ops.attn_block_fwd(_bf(_randn((Bq, S_, Ew), 32)), None, None, _bf(_randn((3 * Ew, Ew), 33, 0.1)), 0.1 * _randn((3 * Ew,), 34),
                                   _bf(_randn((Ew, Ew), 35, 0.1)), 0.1 * _randn((Ew,), 36), 1 + 0.1 * _randn((Ew,), 37), 0.1 * _randn((Ew,), 38),
                                   extras=(S_ == 31))
            for k, v in o.items():
                out[f"attn_fwd_w{Ew}_S{S_}_" + k] = v
    # --- the whole model at the bench shape, B = 4096: forward, loss, backward ---
    # (bf16, and fp16: the same kernels compiled for IEEE half -- elem.h -- with their own counted waits in the object code)
    cfg = orc.MiViTConfig(embedding="linear", patch_size=64, embed_dim=128, num_heads=4, hidden_dim=256, num_layers=4)
    xs = 0.2 + 0.06 * _randn((4096, 32, 64, 64), 41)
    ls = torch.rand(4096, 1, generator=torch.Generator(device="cuda").manual_seed(42), device="cuda")
    for prec in ("bf16", "fp16"):
        m = build_product_model(cfg, prec, orc.closed_form_params(cfg))
        o = m(xs)
        (F.mse_loss(o, ls) * 256.0).backward()
        out[f"model_{prec}_out"] = o.detach()
        for k, p in m.named_parameters():
            out[f"model_{prec}_grad_" + k] = p.grad.detach().clone()
        torch.cuda.synchronize()
        del m
    # the 64-wide model (fused blocks of width 64; wave-stream K = 192 data gradient, wgrad_small) at the Framerate shape
    cfg3 = orc.MiViTConfig(embedding="linear", patch_size=13, embed_dim=64, num_heads=4, hidden_dim=128, num_layers=2)
    m = build_product_model(cfg3, "bf16", orc.closed_form_params(cfg3))
    x3 = 0.2 + 0.06 * _randn((4096, 30, 13, 13), 43)
    o = m(x3)
    F.mse_loss(o, ls).backward()
    out["model_c3_out"] = o.detach()
    for k, p in m.named_parameters():
        out["model_c3_grad_" + k] = p.grad.detach().clone()
    torch.cuda.synchronize()
    del m
    return out


def test_counted_waits_equal_blanket_waits_bitwise(monkeypatch):
    from moleculardiffusion_mivit_amd import _native as N
    counted = _run_everything()
    strict_lib = N.load_strict()
    monkeypatch.setattr(N, "lib", strict_lib)
    strict = _run_everything()
    monkeypatch.undo()
    assert counted.keys() == strict.keys()
    bad = [k for k in counted if not torch.equal(counted[k], strict[k])]
    assert not bad, bad
    assert all(torch.isfinite(v.float()).all() for v in counted.values())
