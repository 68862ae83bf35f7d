"""Operator-level parity: each HIP kernel (through the C-ABI via ops.py) against the oracle's torch-fp32 arithmetic.

Tolerances: fp32 mode 1e-4 relative (north_star); bf16 mode 3e-2 (bf16 operands, fp32 accumulate).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-4, torch.bfloat16: 3e-2, torch.float16: 4e-3}


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a.detach() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-30))


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(264, 128, 128), (240, 64, 81), (7, 1, 128), (1, 3, 5), (300, 384, 128),
                                   (257, 256, 1024), (1000, 128, 4096), (129, 130, 131), (64, 512, 25)])
@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_linear_fwd_bwd(dtype, M, N, K, act):
    from moleculardiffusion_mivit_amd import ops
    if act != 0 and K > 1024:
        pytest.skip("activation variants covered at smaller K")
    x = _mk((M, K), 1)
    W = _mk((N, K), 2, 1.0 / math.sqrt(K))
    b = _mk((N,), 3, 0.1)
    dy = _mk((M, N), 4)
    # oracle arithmetic on the values the kernel actually sees (bf16-rounded activations in bf16 mode)
    xr = x.to(dtype).float().clone().requires_grad_(True)
    # (weights are rounded to the compute type on their way into LDS: give the checker the same values, so
    #  ReLU masks agree except for fp32 summation-order noise)
    Wr = W.to(dtype).float().clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    actf = [lambda t: t, F.relu, F.leaky_relu, F.gelu][act]
    yr = actf(F.linear(xr, Wr, br))
    yr.backward(dy.to(dtype).float())

    xg = x.detach().to(dtype).cuda().requires_grad_(True)
    Wg = W.to(dtype).float().cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True)
    yg = ops.linear(xg, Wg, bg, act=act)
    yg.backward(dy.to(dtype).cuda())
    torch.cuda.synchronize()
    tol = TOL[dtype]
    assert _rel(yg.float(), yr) < tol
    assert _rel(xg.grad.float(), xr.grad) < tol * (3 if dtype != torch.float32 else 1)
    assert _rel(Wg.grad, Wr.grad) < tol * (3 if dtype != torch.float32 else 1)
    assert _rel(bg.grad, br.grad) < tol * (2 if dtype != torch.float32 else 1)


def test_linear_exact_integers():
    """Asymmetric small-integer operands: any fragment / layout mix-up shows as an exact mismatch."""
    from moleculardiffusion_mivit_amd import ops
    for dtype in (torch.float32, torch.bfloat16):
        M, N, K = 96, 80, 64
        x = ((torch.arange(M * K).reshape(M, K) * 7 + 3) % 5 - 2).float()
        W = ((torch.arange(N * K).reshape(N, K) * 11 + 1) % 7 - 3).float()
        dy = ((torch.arange(M * N).reshape(M, N) * 5 + 2) % 3 - 1).float()
        xr, Wr = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
        (xr @ Wr.t()).backward(dy)
        xg, Wg = x.to(dtype).cuda().requires_grad_(True), W.cuda().requires_grad_(True)
        y = ops.linear(xg, Wg, None)
        y.backward(dy.to(dtype).cuda())
        assert torch.equal(y.float().cpu(), x @ W.t())
        assert torch.equal(xg.grad.float().cpu(), xr.grad)
        assert torch.equal(Wg.grad.cpu(), Wr.grad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,E", [(264, 128), (5, 32), (1000, 64), (130, 512), (33, 96), (17, 1024)])
def test_layernorm(dtype, M, E):
    from moleculardiffusion_mivit_amd import ops
    x = _mk((M, E), 5) * 2 + 0.3
    w = 1 + 0.2 * _mk((E,), 6)
    b = 0.1 * _mk((E,), 7)
    dy = _mk((M, E), 8)
    xr = x.to(dtype).float().clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.layer_norm(xr, (E,), wr, br).backward(dy.to(dtype).float())
    xg = x.detach().to(dtype).cuda().requires_grad_(True)
    wg, bg = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    yg = ops.layer_norm(xg, wg, bg)
    yg.backward(dy.to(dtype).cuda())
    tol = TOL[dtype]
    assert _rel(yg.float(), F.layer_norm(xr, (E,), wr, br)) < tol
    assert _rel(xg.grad.float(), xr.grad) < tol
    assert _rel(wg.grad, wr.grad) < tol
    assert _rel(bg.grad, br.grad) < tol


def _attn_ref(qkv, H):
    B, S, E3 = qkv.shape
    E = E3 // 3
    Dh = E // H
    q, k, v = [t.reshape(B, S, H, Dh).permute(0, 2, 1, 3) for t in qkv.split(E, dim=-1)]
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(Dh), dim=-1)
    return (a @ v).permute(0, 2, 1, 3).reshape(B, S, E)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,S,H,Dh", [(8, 33, 4, 32), (4, 31, 4, 16), (2, 65, 8, 64), (3, 7, 2, 16), (2, 16, 2, 32),
                                      (2, 17, 4, 16), (1, 1, 1, 16), (2, 61, 4, 16), (2, 48, 8, 16), (1, 80, 2, 64),
                                      (2, 64, 2, 32), (1, 65, 4, 32), (3, 49, 4, 32), (2, 32, 4, 16), (1, 96, 2, 32),
                                      (1, 112, 2, 32), (2, 128, 2, 16), (1, 100, 3, 16), (5, 65, 8, 64), (1, 113, 2, 32),
                                      (2, 128, 4, 32), (1, 128, 2, 64), (1, 97, 2, 64)])
def test_attention(dtype, B, S, H, Dh):
    from moleculardiffusion_mivit_amd import ops, _native as N
    code = {torch.float32: N.F32, torch.bfloat16: N.BF16, torch.float16: N.F16}[dtype]
    if S > N.lib.mivit_attention_max_seq(code, Dh):
        pytest.skip("sequence length beyond what this precision's attention kernels hold on chip")
    E = H * Dh
    qkv = _mk((B, S, 3 * E), 9)
    do = _mk((B, S, E), 10)
    qr = qkv.to(dtype).float().clone().requires_grad_(True)
    _attn_ref(qr, H).backward(do.to(dtype).float())
    qg = qkv.detach().to(dtype).cuda().requires_grad_(True)
    og = ops.attention(qg, H)
    og.backward(do.to(dtype).cuda())
    tol = TOL[dtype]
    assert _rel(og.float(), _attn_ref(qr, H)) < tol
    assert _rel(qg.grad.float(), qr.grad) < tol * (2 if dtype != torch.float32 else 1)


def test_attention_exact_uniform():
    """q = 0 makes the softmax exactly uniform: ctx must be the mean of v over tokens (checks padding masks)."""
    from moleculardiffusion_mivit_amd import ops
    B, S, H, Dh = 2, 33, 4, 32
    E = H * Dh
    qkv = _mk((B, S, 3 * E), 11)
    qkv[..., :E] = 0
    out = ops.attention(qkv.cuda(), H).cpu()
    ref = qkv[..., 2 * E:].mean(dim=1, keepdim=True).expand(B, S, E)
    assert _rel(out, ref) < 1e-5


@pytest.mark.parametrize("M,K,E", [(256, 4096, 128), (1000, 256, 128), (130, 1024, 256), (4096, 4096, 128), (333, 16384, 512)])
def test_embed_streaming_kernels_exact(M, K, E):
    """LDS-DMA embedding kernels (csrc/embed.hip) on small-integer data: bf16 represents every operand and fp32
    every partial sum exactly, so any swizzle / fragment / ring-slot mix-up is an exact mismatch."""
    import ctypes
    from moleculardiffusion_mivit_amd import _native as N
    g = torch.Generator().manual_seed(M + K)
    x = torch.randint(-3, 4, (M, K), generator=g).float()
    W = torch.randint(-2, 3, (E, K), generator=g).float()
    b = torch.randint(-4, 5, (E,), generator=g).float()
    dy = torch.randint(-2, 3, (M, E), generator=g).float()
    xg, Wg, bg, dyg = x.cuda(), W.bfloat16().cuda(), b.cuda(), dy.bfloat16().cuda()
    y = torch.empty(M, E, dtype=torch.bfloat16, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    N.check(N.lib.mivit_embed_fwd_bf16(p(xg), p(Wg), p(bg), M, K, E, p(y), st), "embed_fwd")
    ref = x.double() @ W.double().t() + b.double()
    assert float(ref.abs().max()) < 2 ** 24           # exactly representable sums
    assert torch.equal(y.float().cpu(), ref.float().bfloat16().float())
    ws = torch.empty(max(N.lib.mivit_embed_wgrad_bf16_workspace_bytes(M, K, E), 16), dtype=torch.uint8, device="cuda")
    dW = torch.empty(E, K, device="cuda")
    N.check(N.lib.mivit_embed_wgrad_bf16(p(dyg), p(xg), M, K, E, p(dW), p(ws), ws.numel(), st), "embed_wgrad")
    assert torch.equal(dW.cpu(), (dy.double().t() @ x.double()).float())


def test_embed_streaming_kernels_reject_odd_shapes():
    import ctypes
    from moleculardiffusion_mivit_amd import _native as N
    z = torch.zeros(16, device="cuda")
    p = ctypes.c_void_p(z.data_ptr())
    assert N.lib.mivit_embed_fwd_bf16(p, p, p, 240, 81, 64, p, None) == 3      # reference shape: general GEMM instead
    assert N.lib.mivit_embed_wgrad_bf16_workspace_bytes(240, 81, 64) == 0


def _ints(shape, lo, hi, seed):
    return torch.randint(lo, hi + 1, shape, generator=torch.Generator().manual_seed(seed)).float()


@pytest.mark.parametrize("M,N,K", [(264, 128, 128), (1000, 384, 128), (4130, 256, 128), (777, 128, 256),
                                   # 64-wide models (the reference's shipped size): wave-stream only
                                   (300, 64, 64), (1000, 192, 64), (515, 128, 64), (700, 64, 128)])
@pytest.mark.parametrize("variant", ["plain", "relu_preact", "resid", "resid_ln"])
@pytest.mark.parametrize("family", ["rowstream", "wavestream"])
def test_rowstream_forward_exact(M, N, K, variant, family):
    """Row-stream / wave-stream GEMMs (csrc/rowstream.hip, wavestream.hip), forward, on small integers: exact up to the
    final bf16 rounding."""
    import ctypes
    from moleculardiffusion_mivit_amd import _native as N_
    entry = getattr(N_.lib, f"mivit_{family}_fwd")
    if family == "rowstream" and (K == 64 or N % 128):
        pytest.skip("64-wide shapes are wave-stream only")
    if variant == "resid_ln" and N not in (64, 128):
        pytest.skip("fused LayerNorm needs the slice to be the whole row")
    x, W, b = _ints((M, K), -2, 2, 1), _ints((N, K), -2, 2, 2), _ints((N,), -3, 3, 3)
    r = _ints((M, N), -4, 4, 4)
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())   # noqa: E731
    xg, Wg, bg, rg = x.bfloat16().cuda(), W.bfloat16().cuda(), b.cuda(), r.bfloat16().cuda()
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    pre = torch.empty_like(y) if variant == "relu_preact" else None
    has_r = variant in ("resid", "resid_ln")
    ln = variant == "resid_ln"
    gam = (1 + 0.1 * torch.randn(N, generator=torch.Generator().manual_seed(5))).cuda() if ln else None
    bet = (0.1 * torch.randn(N, generator=torch.Generator().manual_seed(6))).cuda() if ln else None
    lno = torch.empty_like(y) if ln else None
    mean = torch.empty(M, device="cuda") if ln else None
    rstd = torch.empty(M, device="cuda") if ln else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    N_.check(entry(p(xg), K, p(Wg), p(bg), M, N, K, 1 if variant == "relu_preact" else 0,
                   p(rg) if has_r else None, N, p(y), N, p(pre), p(gam), p(bet), p(lno), p(mean), p(rstd), st), family + "_fwd")
    u = x.double() @ W.double().t() + b.double()
    ref = torch.relu(u) if variant == "relu_preact" else u
    if has_r:
        ref = ref + r.double()
    assert torch.equal(y.float().cpu(), ref.float().bfloat16().float())
    if pre is not None:
        assert torch.equal(pre.float().cpu(), u.float().bfloat16().float())
    if ln:
        z = ref.float().bfloat16().float()
        want = torch.nn.functional.layer_norm(z, (N,), gam.cpu(), bet.cpu())
        assert float((lno.float().cpu() - want).abs().max()) < 2e-2 * float(want.abs().max())
        assert float((mean.cpu() - z.mean(-1)).abs().max()) < 1e-4
        assert float((rstd.cpu() - torch.rsqrt(z.var(-1, unbiased=False) + 1e-5)).abs().max()) < 1e-3


@pytest.mark.parametrize("M,N,K", [(264, 128, 128), (1000, 128, 256), (4130, 256, 128), (900, 384, 128),
                                   (300, 64, 64), (1000, 64, 192), (515, 128, 64), (700, 64, 128),
                                   (1000, 192, 64), (4133, 192, 64)])      # q|k|v data gradient of the 64-wide models
@pytest.mark.parametrize("variant", ["plain", "dact_relu", "dres"])
@pytest.mark.parametrize("family", ["rowstream", "wavestream"])
def test_rowstream_dgrad_exact(M, N, K, variant, family):
    """dx[M,K] = dy[M,N] @ W[N,K] (* relu'(saved)) (+ dres): W consumed through transposed LDS reads."""
    import ctypes
    from moleculardiffusion_mivit_amd import _native as N_
    if family == "wavestream" and N == 384:
        pytest.skip("contraction 384 stays on the row-stream kernel")
    if family == "rowstream" and (N == 64 or K % 128):
        pytest.skip("64-wide shapes are wave-stream only")
    entry = getattr(N_.lib, f"mivit_{family}_dgrad")
    dy, W = _ints((M, N), -2, 2, 7), _ints((N, K), -2, 2, 8)
    saved, dres = _ints((M, K), -1, 2, 9), _ints((M, K), -4, 4, 10)
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())   # noqa: E731
    dyg, Wg, sg, rg = dy.bfloat16().cuda(), W.bfloat16().cuda(), saved.bfloat16().cuda(), dres.bfloat16().cuda()
    dx = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    N_.check(entry(p(dyg), N, p(Wg), M, N, K, 1 if variant == "dact_relu" else 0, p(sg) if variant == "dact_relu" else None, K,
                   p(rg) if variant == "dres" else None, K, p(dx), K, st), family + "_dgrad")
    ref = dy.double() @ W.double()
    if variant == "dact_relu":
        ref = ref * (saved > 0).double()
    if variant == "dres":
        ref = ref + dres.double()
    assert torch.equal(dx.float().cpu(), ref.float().bfloat16().float())


@pytest.mark.parametrize("M,N,K", [(264, 128, 128), (1000, 384, 128), (4130, 128, 256), (70000, 256, 128),
                                   (900, 512, 512), (2100, 1536, 512)])      # wide layers: two ring slots, two workgroups / CU
def test_wgrad_dma_exact(M, N, K):
    import ctypes
    from moleculardiffusion_mivit_amd import _native as N_
    dy, x = _ints((M, N), -2, 2, 11), _ints((M, K), -2, 2, 12)
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    dyg, xg = dy.bfloat16().cuda(), x.bfloat16().cuda()
    ws = torch.empty(max(N_.lib.mivit_wgrad_bf16_workspace_bytes(M, N, K), 16), dtype=torch.uint8, device="cuda")
    dW = torch.empty(N, K, device="cuda")
    db = torch.empty(N, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    N_.check(N_.lib.mivit_wgrad_bf16(p(dyg), N, p(xg), K, M, N, K, p(dW), p(db), p(ws), ws.numel(), st), "wgrad_bf16")
    assert torch.equal(dW.cpu(), (dy.double().t() @ x.double()).float())
    assert torch.equal(db.cpu(), dy.double().sum(0).float())


GEMM_DMA_TILES = [0, 19, 20, 21, 22, 23, 27, 30, 31, 32, 33, 37]     # csrc/gemm_dma.hip launch_variant: default, first generation, transposed-product tiles, persistent


@pytest.fixture
def gemm_dma_tile(request):
    from moleculardiffusion_mivit_amd import _native as N_
    old = N_.lib.mivit_gemm_dma_set_variant(request.param)
    yield request.param
    N_.lib.mivit_gemm_dma_set_variant(old)


@pytest.mark.parametrize("gemm_dma_tile", GEMM_DMA_TILES, indirect=True)
@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (1000, 512, 512), (4130, 1536, 512), (777, 512, 1024), (300, 1024, 576)])
@pytest.mark.parametrize("variant", ["plain", "relu_preact", "resid", "gelu"])
def test_gemm_dma_forward_exact(M, N, K, variant, gemm_dma_tile):
    """Wide-layer LDS-DMA GEMM (csrc/gemm_dma.hip), forward, on small integers: exact up to the final bf16 rounding
    (every tile variant of the sizing sweep; ragged last row tile; K = 576 is not a multiple of 64 but of 32)."""
    import ctypes
    from moleculardiffusion_mivit_amd import _native as N_
    assert N_.lib.mivit_gemm_dma_supported(M, N, K, 0)
    x, W, b = _ints((M, K), -2, 2, 1), _ints((N, K), -2, 2, 2), _ints((N,), -3, 3, 3)
    r = _ints((M, N), -4, 4, 4)
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())   # noqa: E731
    xg, Wg, bg, rg = x.bfloat16().cuda(), W.bfloat16().cuda(), b.cuda(), r.bfloat16().cuda()
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    pre = torch.empty_like(y) if variant in ("relu_preact", "gelu") else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    act = {"relu_preact": 1, "gelu": 3}.get(variant, 0)
    N_.check(N_.lib.mivit_gemm_dma_fwd(p(xg), K, p(Wg), p(bg), M, N, K, act,
                                       p(rg) if variant == "resid" else None, N, p(y), N, p(pre), st), "gemm_dma_fwd")
    u = x.double() @ W.double().t() + b.double()
    ref = torch.relu(u) if variant == "relu_preact" else u
    if variant == "resid":
        ref = ref + r.double()
    if variant == "gelu":
        assert (y.float().cpu() - torch.nn.functional.gelu(u.float())).abs().max() <= 2e-2 * u.abs().max()
    else:
        assert torch.equal(y.float().cpu(), ref.float().bfloat16().float())
    if pre is not None:
        assert torch.equal(pre.float().cpu(), u.float().bfloat16().float())


@pytest.mark.parametrize("gemm_dma_tile", GEMM_DMA_TILES, indirect=True)
@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (1000, 512, 512), (4130, 1536, 512), (777, 1024, 512), (300, 576, 1024)])
@pytest.mark.parametrize("variant", ["plain", "dact_relu", "dres"])
def test_gemm_dma_dgrad_exact(M, N, K, variant, gemm_dma_tile):
    """dx[M,K] = dy[M,N] @ W[N,K] (* relu'(saved)) (+ dres): W tile consumed through transposed LDS reads."""
    import ctypes
    from moleculardiffusion_mivit_amd import _native as N_
    assert N_.lib.mivit_gemm_dma_supported(M, N, K, 1)
    dy, W = _ints((M, N), -2, 2, 7), _ints((N, K), -2, 2, 8)
    saved, dres = _ints((M, K), -1, 2, 9), _ints((M, K), -4, 4, 10)
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())   # noqa: E731
    dyg, Wg, sg, rg = dy.bfloat16().cuda(), W.bfloat16().cuda(), saved.bfloat16().cuda(), dres.bfloat16().cuda()
    dx = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    N_.check(N_.lib.mivit_gemm_dma_dgrad(p(dyg), N, p(Wg), M, N, K, 1 if variant == "dact_relu" else 0,
                                         p(sg) if variant == "dact_relu" else None, K,
                                         p(rg) if variant == "dres" else None, K, p(dx), K, st), "gemm_dma_dgrad")
    ref = dy.double() @ W.double()
    if variant == "dact_relu":
        ref = ref * (saved > 0).double()
    if variant == "dres":
        ref = ref + dres.double()
    assert torch.equal(dx.float().cpu(), ref.float().bfloat16().float())


@pytest.mark.parametrize("M,N,K", [(256, 64, 64), (1000, 128, 64), (4131, 192, 64), (70000, 64, 128), (300, 192, 64)])
def test_wgrad_small_exact(M, N, K):
    """Narrow-layer weight gradient (csrc/wgrad_small.hip) on small integers: every partial sum is exact in fp32."""
    import ctypes
    from moleculardiffusion_mivit_amd import _native as N_
    dy, x = _ints((M, N), -2, 2, 21), _ints((M, K), -2, 2, 22)
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    dyg, xg = dy.bfloat16().cuda(), x.bfloat16().cuda()
    ws = torch.empty(N_.lib.mivit_wgrad_small_workspace_bytes(M, N, K), dtype=torch.uint8, device="cuda")
    dW = torch.empty(N, K, device="cuda")
    db = torch.empty(N, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    N_.check(N_.lib.mivit_wgrad_small(p(dyg), N, p(xg), K, M, N, K, p(dW), p(db), p(ws), ws.numel(), st), "wgrad_small")
    assert torch.equal(dW.cpu(), (dy.double().t() @ x.double()).float())
    assert torch.equal(db.cpu(), dy.double().sum(0).float())
    dW.zero_()
    N_.check(N_.lib.mivit_wgrad_small(p(dyg), N, p(xg), K, M, N, K, p(dW), None, p(ws), ws.numel(), st), "wgrad_small")
    assert torch.equal(dW.cpu(), (dy.double().t() @ x.double()).float())


# ---- DeepResNet staged entry points (synchronised BatchNorm ABI) with a world of one ---------------------------------
@pytest.mark.parametrize("dtype_name,P,n", [("fp32", 9, 12), ("bf16", 9, 40), ("bf16", 16, 6)])
def test_deepresnet_staged_entry_points_match_the_single_call(dtype_name, P, n):
    """mivit_deepresnet_train_fwd_stage / _bwd_stage called stage by stage with *global_count = this rank's own count and no
    exchange in between must reproduce mivit_deepresnet_train_fwd / _bwd (same kernels; the statistics go through one fp64
    row instead of the fp32 partial chain, so not bitwise: 1e-5 of the largest value).  Also pins the workspace layout
    introspection: the raw convolution outputs found at the reported offsets have the reported shapes and are finite."""
    import ctypes
    from moleculardiffusion_mivit_amd import _native as N
    from moleculardiffusion_mivit_amd.helpers.models import DeepResNetEmbedding
    E = 32
    torch.manual_seed(P * 100 + n)
    emb = DeepResNetEmbedding(P, E).cuda()
    x = (torch.rand(n, P, P, device="cuda") * 1.5 - 0.25).contiguous()
    dtok = torch.randn(n, E, device="cuda")
    params = []
    for conv, bn in emb._conv_bn_pairs():
        params += [conv.weight.detach(), bn.weight.detach(), bn.bias.detach()]
    params += [emb.fc.weight.detach(), emb.fc.bias.detach()]
    code = N.F32 if dtype_name == "fp32" else N.BF16
    nbytes = N.lib.mivit_deepresnet_train_workspace_bytes(code, n, P, E)
    assert nbytes > 0
    off = (ctypes.c_size_t * 16)()
    N.check(N.lib.mivit_deepresnet_train_workspace_layout(code, n, P, E, ctypes.addressof(off)), "layout")
    assert off[15] == nbytes and list(off[:15]) == sorted(off[:15])
    s = torch.cuda.current_stream().cuda_stream

    def run(staged):
        prm, gr = N.DeepResNetParams(), N.DeepResNetGrads()
        grads = [torch.zeros_like(t) for t in params]
        for i in range(7):
            w, g, b = params[3 * i:3 * i + 3]
            prm.conv[i] = N.ConvBn(w.data_ptr(), g.data_ptr(), b.data_ptr(), None, None)
            gr.conv[i] = N.ConvBnGrad(*[t.data_ptr() for t in grads[3 * i:3 * i + 3]])
        prm.fc_weight, prm.fc_bias = params[21].data_ptr(), params[22].data_ptr()
        gr.fc_weight, gr.fc_bias = grads[21].data_ptr(), grads[22].data_ptr()
        ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        tokens = torch.empty(n, E, device="cuda")
        if staged:
            count = torch.full((1,), float(n * P * P), dtype=torch.float64, device="cuda")
            stats = torch.zeros(2, 3, 128, dtype=torch.float64, device="cuda")
            for st in range(6):
                N.check(N.lib.mivit_deepresnet_train_fwd_stage(code, ctypes.addressof(prm), x.data_ptr(), n, P, E, 0.1, 1e-5,
                                                               tokens.data_ptr(), ws.data_ptr(), nbytes, st, count.data_ptr(),
                                                               stats.data_ptr(), s), "fwd_stage")
            for st in range(6):
                N.check(N.lib.mivit_deepresnet_train_bwd_stage(code, ctypes.addressof(prm), x.data_ptr(), dtok.data_ptr(), n, P, E, 1e-5,
                                                               ctypes.addressof(gr), ws.data_ptr(), nbytes, st, count.data_ptr(),
                                                               stats.data_ptr(), s), "bwd_stage")
        else:
            N.check(N.lib.mivit_deepresnet_train_fwd(code, ctypes.addressof(prm), x.data_ptr(), n, P, E, 0.1, 1e-5, tokens.data_ptr(),
                                                     ws.data_ptr(), nbytes, s), "fwd")
            N.check(N.lib.mivit_deepresnet_train_bwd(code, ctypes.addressof(prm), x.data_ptr(), dtok.data_ptr(), n, P, E, 1e-5,
                                                     ctypes.addressof(gr), ws.data_ptr(), nbytes, s), "bwd")
        torch.cuda.synchronize()
        return tokens, grads, ws

    t_a, g_a, ws = run(False)
    t_b, g_b, _ = run(True)
    assert float((t_a - t_b).abs().max()) <= 1e-5 * float(t_a.abs().max())
    gmax = max(float(g.abs().max()) for g in g_a)
    for ga, gb in zip(g_a, g_b):
        assert float((ga - gb).abs().max()) <= 1e-5 * gmax + (3e-3 * gmax if dtype_name == "bf16" else 0.0)
    es = 4 if dtype_name == "fp32" else 2
    y0 = ws[off[0]:off[0] + n * P * P * 32 * es].view(torch.float32 if es == 4 else torch.bfloat16)
    assert y0.numel() == n * P * P * 32 and bool(torch.isfinite(y0.float()).all()) and float(y0.float().abs().max()) > 0
    with pytest.raises(N.MivitError):
        N.check(N.lib.mivit_deepresnet_train_fwd_stage(code, None, x.data_ptr(), n, P, E, 0.1, 1e-5, t_a.data_ptr(), ws.data_ptr(),
                                                       nbytes, 9, None, None, s), "fwd_stage")
