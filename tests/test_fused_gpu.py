"""Fused encoder-layer block kernels (csrc/fused_fwd.hip, csrc/fused_bwd.hip) against plain torch fp32 arithmetic
of reference helpers/models.py:33-59 (attention), :72-77 (feed-forward), :97-108 (post-norm wiring), evaluated on the
bf16-rounded operands the kernels see.  Tolerance 3e-2 relative (bf16 operands, fp32 accumulate) on every output;
small-integer cases are exact."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

E, FH, H = 128, 256, 4


@pytest.fixture(autouse=True, params=[128, 64], ids=["w128", "w64"])
def _layer_width(request):
    """every test of this file runs at both widths the fused blocks are compiled for (elem.h: E = 128 / F = 256 / head dim 32
    and the reference's shipped E = 64 / F = 128 / head dim 16, Experiments/Framerate/trainSettingsFramerate.py:42-47)"""
    global E, FH
    old = (E, FH)
    E, FH = request.param, 2 * request.param
    yield request.param
    E, FH = old


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _bf(t):
    return t.to(torch.bfloat16)


ACTS = {1: F.relu, 2: F.leaky_relu, 3: F.gelu}


def _ln_hat(z):
    mu = z.mean(-1, keepdim=True)
    var = ((z - mu) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + 1e-5)
    return (z - mu) * rstd, mu.squeeze(-1), rstd.squeeze(-1)


@pytest.mark.parametrize("M", [1, 33, 264, 1000, 4097])
@pytest.mark.parametrize("act", [1, 2, 3])
@pytest.mark.parametrize("affine", [False, True])
def test_mlp_block_fwd(M, act, affine):
    from moleculardiffusion_mivit_amd import ops
    n_in = _bf(_mk((M, E), 1))
    gi, bi = (1.0 + 0.3 * _mk((E,), 2), 0.2 * _mk((E,), 3)) if affine else (None, None)
    W1, b1 = _bf(_mk((FH, E), 4, 1 / math.sqrt(E))), 0.1 * _mk((FH,), 5)
    W2, b2 = _bf(_mk((E, FH), 6, 1 / math.sqrt(FH))), 0.1 * _mk((E,), 7)
    go, bo = 1.0 + 0.3 * _mk((E,), 8), 0.2 * _mk((E,), 9)
    x = n_in.float() * gi + bi if affine else n_in.float()
    xb = _bf(x).float()                       # the kernel's x: rounded to bf16 once (MFMA operand and residual alike)
    u = F.linear(xb, W1.float(), b1)
    h = ACTS[act](u)
    z = xb + F.linear(_bf(h).float(), W2.float(), b2)
    nh, mu, rstd = _ln_hat(z)
    out = ops.mlp_block_fwd(n_in.cuda(), gi.cuda() if affine else None, bi.cuda() if affine else None, W1.cuda(), b1.cuda(),
                            W2.cuda(), b2.cuda(), go.cuda(), bo.cuda(), act=act, extras=True)
    torch.cuda.synchronize()
    assert _rel(out["u"].float(), u) < 2e-2
    assert _rel(out["h"].float(), h) < 2e-2
    assert _rel(out["z"].float(), z) < 2e-2
    assert _rel(out["n"].float(), nh) < 3e-2
    assert _rel(out["x"].float(), nh * go + bo) < 3e-2
    assert _rel(out["rstd"], rstd) < 1e-2
    assert float((out["mean"].cpu() - mu).abs().max()) < 2e-2
    lean = ops.mlp_block_fwd(n_in.cuda(), gi.cuda() if affine else None, bi.cuda() if affine else None, W1.cuda(), b1.cuda(),
                             W2.cuda(), b2.cuda(), go.cuda(), bo.cuda(), act=act)
    assert torch.equal(lean["n"], out["n"]) and torch.equal(lean["rstd"], out["rstd"])


def test_mlp_block_fwd_exact_integers():
    """Small-integer operands: every product and sum is exact in bf16 / fp32, so a wrong fragment layout or a wrong
    permutation of the hidden index shows as an exact mismatch in h and z."""
    from moleculardiffusion_mivit_amd import ops
    M = 200
    n_in = ((torch.arange(M * E).reshape(M, E) * 7 + 3) % 5 - 2).float()
    W1 = (((torch.arange(FH * E).reshape(FH, E) * 11 + 1) % 23) == 0).float() * (((torch.arange(FH * E).reshape(FH, E)) % 3) - 1.0)
    W2 = (((torch.arange(E * FH).reshape(E, FH) * 5 + 2) % 29) == 0).float() * (((torch.arange(E * FH).reshape(E, FH)) % 5) - 2.0)
    b1 = ((torch.arange(FH) % 7) - 3).float()
    b2 = ((torch.arange(E) % 5) - 2).float()
    u = F.linear(n_in, W1, b1)
    h = F.relu(u)
    z = n_in + F.linear(h, W2, b2)
    assert float(u.abs().max()) <= 256 and float(z.abs().max()) <= 256        # exactly representable in bf16
    ones, zeros = torch.ones(E), torch.zeros(E)
    out = ops.mlp_block_fwd(_bf(n_in).cuda(), None, None, _bf(W1).cuda(), b1.cuda(), _bf(W2).cuda(), b2.cuda(), ones.cuda(),
                            zeros.cuda(), act=1, extras=True)
    torch.cuda.synchronize()
    assert torch.equal(out["u"].float().cpu(), u)
    assert torch.equal(out["h"].float().cpu(), h)
    assert torch.equal(out["z"].float().cpu(), z)


def _attn_ref(x, xb, Wqkv, bqkv, Wo, bo, S):
    B = x.shape[0]
    qkv = F.linear(xb, Wqkv.float(), bqkv)
    q, k, v = [_bf(t).float().view(B, S, H, E // H).transpose(1, 2) for t in qkv.split(E, dim=-1)]
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(float(E // H)), dim=-1)
    ctx = (_bf(p).float() @ v).transpose(1, 2).reshape(B, S, E)
    z = xb + F.linear(_bf(ctx).float(), Wo.float(), bo)
    return qkv, ctx, z


@pytest.mark.parametrize("B,S", [(1, 1), (3, 7), (2, 16), (5, 17), (9, 31), (64, 33), (3, 48), (2, 61), (2, 64), (300, 33)])
@pytest.mark.parametrize("affine", [False, True])
def test_attn_block_fwd(B, S, affine):
    from moleculardiffusion_mivit_amd import ops
    n_in = _bf(_mk((B, S, E), 11))
    gi, bi = (1.0 + 0.3 * _mk((E,), 12), 0.2 * _mk((E,), 13)) if affine else (None, None)
    Wqkv, bqkv = _bf(_mk((3 * E, E), 14, 1.5 / math.sqrt(E))), 0.1 * _mk((3 * E,), 15)
    Wo, bo = _bf(_mk((E, E), 16, 1 / math.sqrt(E))), 0.1 * _mk((E,), 17)
    go, bo2 = 1.0 + 0.3 * _mk((E,), 18), 0.2 * _mk((E,), 19)
    x = n_in.float() * gi + bi if affine else n_in.float()
    qkv, ctx, z = _attn_ref(x, _bf(x).float(), Wqkv, bqkv, Wo, bo, S)
    nh, mu, rstd = _ln_hat(z)
    dv = lambda t: None if t is None else t.cuda()
    out = ops.attn_block_fwd(n_in.cuda(), dv(gi), dv(bi), Wqkv.cuda(), bqkv.cuda(), Wo.cuda(), bo.cuda(), go.cuda(), bo2.cuda(),
                             extras=True)
    torch.cuda.synchronize()
    assert _rel(out["qkv"].float(), qkv) < 2e-2
    assert _rel(out["ctx"].float(), ctx) < 3e-2
    assert _rel(out["z"].float(), z) < 3e-2
    assert _rel(out["n"].float(), nh) < 3e-2
    assert _rel(out["x"].float(), nh * go + bo2) < 3e-2
    assert _rel(out["rstd"], rstd) < 2e-2
    lean = ops.attn_block_fwd(n_in.cuda(), dv(gi), dv(bi), Wqkv.cuda(), bqkv.cuda(), Wo.cuda(), bo.cuda(), go.cuda(), bo2.cuda())
    # (two instantiations of the kernel: the compiler may contract different multiply-adds, so not bitwise)
    assert _rel(lean["n"].float(), out["n"].float()) < 1e-2 and _rel(lean["ctx"].float(), out["ctx"].float()) < 1e-2


def test_attn_block_fwd_uniform_softmax_exact():
    """Zero q/k weights -> uniform probabilities 1/S; with S a power of two and small-integer v, ctx is exact."""
    from moleculardiffusion_mivit_amd import ops
    B, S = 4, 32
    n_in = ((torch.arange(B * S * E).reshape(B, S, E) * 5 + 1) % 7 - 3).float()
    Wqkv = torch.zeros(3 * E, E)
    idx = torch.arange(E)
    Wqkv[2 * E + idx, (idx * 37 + 5) % E] = 1.0            # v = a column permutation of x
    Wqkv[2 * E + idx, (idx * 11 + 3) % E] += 2.0
    Wo = torch.zeros(E, E)
    Wo[idx, (idx * 13 + 7) % E] = 1.0
    bqkv, bo = torch.zeros(3 * E), torch.zeros(E)
    v = F.linear(n_in, Wqkv[2 * E:])
    ctx = v.mean(dim=1, keepdim=True).expand(B, S, E)
    z = n_in + F.linear(ctx, Wo)
    out = ops.attn_block_fwd(_bf(n_in).cuda(), None, None, _bf(Wqkv).cuda(), bqkv.cuda(), _bf(Wo).cuda(), bo.cuda(),
                             torch.ones(E).cuda(), torch.zeros(E).cuda(), extras=True)
    torch.cuda.synchronize()
    assert torch.equal(out["qkv"].float().cpu()[..., 2 * E:], v)
    assert _rel(out["ctx"].float(), ctx) < 8e-3           # 1/32 * sum of bf16-exact integers, rounded to bf16 once
    assert _rel(out["z"].float(), z) < 8e-3


def _act_grad(act, u):
    if act == 1:
        return (u > 0).float()
    if act == 2:
        return torch.where(u > 0, torch.ones_like(u), torch.full_like(u, 0.01))
    return 0.5 * (1 + torch.erf(u / math.sqrt(2.0))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)


@pytest.fixture(params=[8, 4])
def mlp_bwd_waves(request):
    """both kernels behind mivit_mlp_block_bwd: hidden units split over 8 waves (default) / 4 waves (csrc/fused_bwd.hip)"""
    from moleculardiffusion_mivit_amd import _native as N
    old = N.lib.mivit_mlp_block_bwd_set_waves(request.param)
    yield request.param
    N.lib.mivit_mlp_block_bwd_set_waves(old)


@pytest.mark.parametrize("M", [1, 31, 264, 1000, 8200])
@pytest.mark.parametrize("act", [1, 2, 3])
def test_mlp_block_bwd(M, act, mlp_bwd_waves):
    """Fused feed-forward backward vs the chain rule of x2 = LN2(x1 + fc2(act(fc1 x1))), x1 = gamma1 * n1 + beta1, written
    out in fp32.  The pre-activations are taken as the kernel forms them (x1 rounded to bf16 times the bf16 weights): a ReLU unit whose pre-activation lies within bf16 rounding of zero would otherwise take the other branch in
    one of the two computations -- the function is discontinuous there."""
    from moleculardiffusion_mivit_amd import ops
    n1 = _bf(_mk((M, E), 21)).float()
    g1, be1 = 1.0 + 0.3 * _mk((E,), 22), 0.2 * _mk((E,), 23)
    W1, b1 = _bf(_mk((FH, E), 24, 1 / math.sqrt(E))).float(), 0.1 * _mk((FH,), 25)
    W2, b2 = _bf(_mk((E, FH), 26, 1 / math.sqrt(FH))).float(), 0.1 * _mk((E,), 27)
    g2, be2 = 1.0 + 0.3 * _mk((E,), 28), 0.2 * _mk((E,), 29)
    dy = _bf(_mk((M, E), 30)).float()
    x1 = n1 * g1 + be1
    u = F.linear(_bf(x1).float(), W1, b1)          # the kernel's recompute: x1 rounded to bf16, fc1's bf16 weights
    h = ACTS[act](u)
    z2 = x1 + F.linear(h, W2, b2)
    nh, _, rstd = _ln_hat(z2)
    gdy = dy * g2
    dz2 = rstd[:, None] * (gdy - gdy.mean(-1, keepdim=True) - nh * (gdy * nh).mean(-1, keepdim=True))
    dh = (dz2 @ W2) * _act_grad(act, u)
    ref = {"dx1": dh @ W1 + dz2, "dW1": dh.t() @ x1, "db1": dh.sum(0), "dW2": dz2.t() @ h, "db2": dz2.sum(0),
           "dgamma2": (dy * nh).sum(0), "dbeta2": dy.sum(0)}
    out = ops.mlp_block_bwd(_bf(dy).cuda(), _bf(nh).cuda(), rstd.cuda(), g2.cuda(), _bf(n1).cuda(), g1.cuda(), be1.cuda(),
                            _bf(W1).cuda(), b1.cuda(), _bf(W2).cuda(), act=act)
    torch.cuda.synchronize()
    for k, r in ref.items():
        assert _rel(out[k].float(), r) < 3e-2, k


def test_mlp_block_bwd_is_deterministic(mlp_bwd_waves):
    from moleculardiffusion_mivit_amd import ops
    M = 5000
    args = [_bf(_mk((M, E), 31)).cuda(), _bf(_mk((M, E), 32)).cuda(), (1 + 0.1 * _mk((M,), 33).abs()).cuda(), (1 + 0.1 * _mk((E,), 34)).cuda(),
            _bf(_mk((M, E), 35)).cuda(), (1 + 0.1 * _mk((E,), 36)).cuda(), (0.1 * _mk((E,), 37)).cuda(),
            _bf(_mk((FH, E), 38, 0.1)).cuda(), (0.1 * _mk((FH,), 39)).cuda(), _bf(_mk((E, FH), 40, 0.1)).cuda()]
    a, b = ops.mlp_block_bwd(*args), ops.mlp_block_bwd(*args)
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("M", [1, 31, 264, 1000, 16500])
def test_attn_out_bwd(M):
    """LayerNorm-1 backward + out-projection backward vs the chain rule in fp32 (reference models.py:57,100-102)."""
    from moleculardiffusion_mivit_amd import ops
    dy = _bf(_mk((M, E), 41)).float()
    nh = _bf(_mk((M, E), 42)).float()
    rstd = 1.0 + 0.2 * _mk((M,), 43).abs()
    g1 = 1.0 + 0.3 * _mk((E,), 44)
    ctx = _bf(_mk((M, E), 45)).float()
    Wo = _bf(_mk((E, E), 46, 1 / math.sqrt(E))).float()
    gdy = dy * g1
    dz1 = rstd[:, None] * (gdy - gdy.mean(-1, keepdim=True) - nh * (gdy * nh).mean(-1, keepdim=True))
    dzb = _bf(dz1).float()                    # the MFMAs consume the bf16 image of dz1
    ref = {"dz1": dz1, "dctx": dzb @ Wo, "dWo": dzb.t() @ ctx, "dbo": dz1.sum(0), "dgamma1": (dy * nh).sum(0), "dbeta1": dy.sum(0)}
    out = ops.attn_out_bwd(_bf(dy).cuda(), _bf(nh).cuda(), rstd.cuda(), g1.cuda(), _bf(ctx).cuda(), _bf(Wo).cuda())
    torch.cuda.synchronize()
    for k, r in ref.items():
        assert _rel(out[k].float(), r) < 2e-2, k
    again = ops.attn_out_bwd(_bf(dy).cuda(), _bf(nh).cuda(), rstd.cuda(), g1.cuda(), _bf(ctx).cuda(), _bf(Wo).cuda())
    for k in out:
        assert torch.equal(out[k], again[k]), k


@pytest.mark.parametrize("M", [1, 31, 264, 1000, 16500])
def test_qkv_bwd(M):
    """q|k|v projection backward (weight, bias and data gradient + residual in one pass, csrc/fused_bwd.hip::qkv_bwd_kernel)
    vs the chain rule of qkv = x Wqkv^T + bqkv in fp32 (reference models.py:42-44,100-102)."""
    from moleculardiffusion_mivit_amd import ops
    dqkv = _bf(_mk((M, 3 * E), 51)).float()
    x = _bf(_mk((M, E), 52)).float()
    W = _bf(_mk((3 * E, E), 53, 1 / math.sqrt(E))).float()
    res = _bf(_mk((M, E), 54)).float()
    ref = {"dx": dqkv @ W + res, "dW": dqkv.t() @ x, "db": dqkv.sum(0)}
    out = ops.qkv_bwd(_bf(dqkv).cuda(), _bf(x).cuda(), _bf(W).cuda(), _bf(res).cuda())
    torch.cuda.synchronize()
    for k, r in ref.items():
        assert _rel(out[k].float(), r) < 1e-2, k
    again = ops.qkv_bwd(_bf(dqkv).cuda(), _bf(x).cuda(), _bf(W).cuda(), _bf(res).cuda())
    for k in out:
        assert torch.equal(out[k], again[k]), k


def test_qkv_bwd_exact_integers():
    """Small-integer operands: every product and partial sum is exact, so a wrong fragment layout or slab index is an exact mismatch."""
    from moleculardiffusion_mivit_amd import ops
    M = 777
    ar = torch.arange(M * 3 * E).reshape(M, 3 * E)
    dqkv = ((ar * 7 + 3) % 5 - 2).float()
    x = ((torch.arange(M * E).reshape(M, E) * 11 + 1) % 3 - 1).float()
    aw = torch.arange(3 * E * E).reshape(3 * E, E)
    W = (((aw * 13 + 5) % 31) == 0).float() * ((aw % 3) - 1.0)
    res = ((torch.arange(M * E).reshape(M, E) * 3 + 2) % 7 - 3).float()
    out = ops.qkv_bwd(_bf(dqkv).cuda(), _bf(x).cuda(), _bf(W).cuda(), _bf(res).cuda())
    torch.cuda.synchronize()
    dx = dqkv @ W + res
    assert float(dx.abs().max()) <= 256
    assert torch.equal(out["dx"].float().cpu(), dx)
    assert torch.equal(out["dW"].cpu(), dqkv.t() @ x)
    assert torch.equal(out["db"].cpu(), dqkv.sum(0))
