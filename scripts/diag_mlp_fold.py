import math, sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from moleculardiffusion_mivit_amd import ops
E, FH = 128, 256
def _randn(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, generator=g, device="cuda") * scale
_bf = lambda t: t.to(torch.bfloat16)
M = 140017
n_in = _bf(_randn((M, E), 1))
gi, bi = 1.0 + 0.3 * _randn((E,), 2), 0.2 * _randn((E,), 3)
W1, b1 = _bf(_randn((FH, E), 4, 1 / math.sqrt(E))), 0.1 * _randn((FH,), 5)
W2, b2 = _bf(_randn((E, FH), 6, 1 / math.sqrt(FH))), 0.1 * _randn((E,), 7)
go, bo = 1.0 + 0.3 * _randn((E,), 8), 0.2 * _randn((E,), 9)
xb = _bf(n_in.float() * gi + bi).float()
u = F.linear(xb, W1.float(), b1)
# the kernel's own arithmetic: n x bf16(W1 * gamma) + (b1 + W1 beta)
Wf = _bf(W1.float() * gi[None, :]).float()
u2 = F.linear(n_in.float(), Wf, b1 + W1.float() @ bi)
for act in (1, 3):
    out = ops.mlp_block_fwd(n_in, gi, bi, W1, b1, W2, b2, go, bo, act=act, extras=True)
    d = (out["u"].float() - u).abs()
    d2 = (out["u"].float() - _bf(u2).float()).abs()
    i = int(d.argmax()); r, c = divmod(i, FH)
    print(f"act {act}: max|u|={float(u.abs().max()):.3f} maxerr vs ref {float(d.max()):.4f} at row {r} col {c} (u={float(u[r,c]):.4f}, kernel={float(out['u'][r,c]):.4f}, folded-ref={float(u2[r,c]):.4f}); "
          f"vs folded arithmetic {float(d2.max()):.4f}; rows with err>0.05: {int((d.max(1).values>0.05).sum())}; mean err {float(d.mean()):.5f}")
    big = (d.max(1).values > 0.05).nonzero().flatten()[:20].tolist()
    print("   rows:", big)
