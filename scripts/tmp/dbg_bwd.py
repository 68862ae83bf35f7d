import math, sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_fused_gpu import _mk, _bf, _ln_hat, ACTS, E, FH
from moleculardiffusion_mivit_amd import ops
for M in (31, 264):
    act = 1
    n1 = _bf(_mk((M, E), 21)).float()
    g1, be1 = 1.0 + 0.3 * _mk((E,), 22), 0.2 * _mk((E,), 23)
    W1, b1 = _bf(_mk((FH, E), 24, 1 / math.sqrt(E))).float(), 0.1 * _mk((FH,), 25)
    W2, b2 = _bf(_mk((E, FH), 26, 1 / math.sqrt(FH))).float(), 0.1 * _mk((E,), 27)
    g2, be2 = 1.0 + 0.3 * _mk((E,), 28), 0.2 * _mk((E,), 29)
    dy = _bf(_mk((M, E), 30)).float()
    x1 = (n1 * g1 + be1).requires_grad_(True)
    W1r, b1r, W2r, b2r, g2r, be2r = [t.clone().requires_grad_(True) for t in (W1, b1, W2, b2, g2, be2)]
    z2 = x1 + F.linear(ACTS[act](F.linear(x1, W1r, b1r)), W2r, b2r)
    nh, _, rstd = _ln_hat(z2)
    (((nh * g2r + be2r) * dy).sum()).backward()
    out = ops.mlp_block_bwd(_bf(dy).cuda(), _bf(nh.detach()).cuda(), rstd.detach().cuda(), g2.cuda(), _bf(n1).cuda(), g1.cuda(),
                            be1.cuda(), _bf(W1).cuda(), b1.cuda(), _bf(W2).cuda(), act=act)
    ref = dict(dx1=x1.grad, dW1=W1r.grad, db1=b1r.grad, dW2=W2r.grad, db2=b2r.grad, dgamma2=g2r.grad, dbeta2=be2r.grad)
    for k, r in ref.items():
        d = (out[k].float().cpu() - r).abs()
        i = int(d.argmax())
        print(M, k, "max err", float(d.max()), "ref max", float(r.abs().max()), "argmax", i, "idx", [i // r.shape[-1], i % r.shape[-1]] if r.dim() == 2 else i,
              "got", float(out[k].float().cpu().flatten()[i]), "ref", float(r.flatten()[i]))
    d = (out["dW1"].cpu() - W1r.grad).abs()
    print("rows with err>0.05*max:", (d.max(dim=1).values > 0.05 * W1r.grad.abs().max()).nonzero().flatten().tolist()[:40])
    d1 = (out["db1"].cpu() - b1r.grad).abs()
    print("db1 bad:", (d1 > 0.05 * b1r.grad.abs().max()).nonzero().flatten().tolist()[:40])
