"""Diagnostic: run the native DeepResNet step stage by stage (world of one) and check the intermediates in the workspace."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculardiffusion_mivit_amd import _native as N
from moleculardiffusion_mivit_amd.helpers.models import DeepResNetEmbedding

P, n = int(sys.argv[1]), int(sys.argv[2])
E = 64
torch.manual_seed(7)
emb = DeepResNetEmbedding(P, E).cuda().train()
x = (torch.rand(n, P, P, device="cuda") * 1.5 - 0.25).contiguous()
wgt = torch.randn(n, E, device="cuda")
pairs = emb._conv_bn_pairs()
params = []
for conv, bn in pairs:
    params += [conv.weight.detach(), bn.weight.detach(), bn.bias.detach()]
params += [emb.fc.weight.detach(), emb.fc.bias.detach()]
prm, gr = N.DeepResNetParams(), N.DeepResNetGrads()
grads = [torch.zeros_like(t) for t in params]
for i in range(7):
    w, g, b = params[3 * i:3 * i + 3]
    prm.conv[i] = N.ConvBn(w.data_ptr(), g.data_ptr(), b.data_ptr(), None, None)
    gr.conv[i] = N.ConvBnGrad(*[t.data_ptr() for t in grads[3 * i:3 * i + 3]])
prm.fc_weight, prm.fc_bias = params[21].data_ptr(), params[22].data_ptr()
gr.fc_weight, gr.fc_bias = grads[21].data_ptr(), grads[22].data_ptr()
code = N.F32
nbytes = N.lib.mivit_deepresnet_train_workspace_bytes(code, n, P, E)
ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
off = (ctypes.c_size_t * 16)()
N.check(N.lib.mivit_deepresnet_train_workspace_layout(code, n, P, E, ctypes.addressof(off)), "layout")
tokens = torch.empty(n, E, device="cuda")
count = torch.full((1,), float(n * P * P), dtype=torch.float64, device="cuda")
stats = torch.zeros(2, 3, 128, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for st in range(6):
    N.check(N.lib.mivit_deepresnet_train_fwd_stage(code, ctypes.addressof(prm), x.data_ptr(), n, P, E, 0.1, 1e-5, tokens.data_ptr(),
                                                   ws.data_ptr(), nbytes, st, count.data_ptr(), stats.data_ptr(), s), "fwd")
torch.cuda.synchronize()
R = n * P * P


def region(i, rows, cols, dt=torch.float32):
    return ws[off[i]:off[i] + rows * cols * 4].view(dt).view(rows, cols)


fco = ws[off[7]:off[7] + 7 * 4 * 128 * 4].view(torch.float32).view(7, 4, 128)
y5, y6 = region(5, R, 128).clone(), region(6, R, 128).clone()
N.check(N.lib.mivit_deepresnet_train_bwd_stage(code, ctypes.addressof(prm), x.data_ptr(), wgt.data_ptr(), n, P, E, 1e-5,
                                               ctypes.addressof(gr), ws.data_ptr(), nbytes, 0, count.data_ptr(), stats.data_ptr(), s),
        "bwd0")
torch.cuda.synchronize()
dpooled = region(10, n, 128)
ref_dp = wgt @ params[21]
print("dpooled err", float((dpooled - ref_dp).abs().max()), "max", float(ref_dp.abs().max()))
act = fco[5, 2] * y5 + fco[5, 3] + fco[6, 2] * y6 + fco[6, 3]
g_ref = (ref_dp / (P * P)).repeat_interleave(P * P, dim=0) * (act > 0)
X1 = region(12, R, 128)
bad = (X1 - g_ref).abs().amax(dim=1)
print("g2 err", float(bad.max()), "rows wrong", int((bad > 1e-7).sum()), "first", (bad > 1e-7).nonzero().flatten()[:10].tolist())
print("y5 unchanged", bool((region(5, R, 128) == y5).all()), "y6 unchanged", bool((region(6, R, 128) == y6).all()))
sums = torch.stack([g_ref.double().sum(0), (g_ref.double() * y5).sum(0), (g_ref.double() * y6).sum(0)])
print("sums err", float((stats[0] - sums).abs().max()), "max", float(sums.abs().max()))
part = ws[off[11]:off[11] + 64 * 3 * 128 * 4].view(torch.float32).view(64, 3, 128)
nbk = (R + 511) // 512
for b in range(nbk):
    ref_b = g_ref[b * 512:(b + 1) * 512].double().sum(0)
    e = float((part[b, 0] - ref_b).abs().max())
    if e > 1e-5:
        print("  block", b, "s1 err", e)

# ---- fp64 torch restatement that keeps its intermediates, then the remaining backward stages one by one
import torch.nn.functional as F
pd = [t.double().requires_grad_(True) for t in params]
xs = x.double().unsqueeze(1)


def bn(y, i):
    return F.batch_norm(y, None, None, pd[3 * i + 1], pd[3 * i + 2], True, 0.1, 1e-5)


ys = [None] * 7
ys[0] = F.conv2d(xs, pd[0], padding=1); a0 = F.relu(bn(ys[0], 0))
ys[1] = F.conv2d(a0, pd[3], padding=1); ys[3] = F.conv2d(a0, pd[9]); a11 = F.relu(bn(ys[1], 1))
ys[2] = F.conv2d(a11, pd[6], padding=1); o1 = F.relu(bn(ys[2], 2) + bn(ys[3], 3))
ys[4] = F.conv2d(o1, pd[12], padding=1); ys[6] = F.conv2d(o1, pd[18]); a21 = F.relu(bn(ys[4], 4))
ys[5] = F.conv2d(a21, pd[15], padding=1); o2 = F.relu(bn(ys[5], 5) + bn(ys[6], 6))
for t in ys:
    t.retain_grad()
tok = o2.mean(dim=(2, 3)) @ pd[21].t() + pd[22]
(tok * wgt.double()).sum().backward()
CO = [32, 64, 64, 64, 128, 128, 128]
bco = ws[off[8]:off[8] + 7 * 3 * 128 * 4].view(torch.float32).view(7, 3, 128)
flips = ((act > 0) != (o2.detach().permute(0, 2, 3, 1).reshape(R, 128) > 0)).sum()
print("relu flips at the top:", int(flips))


def check_dy(i, xbuf):
    C = CO[i]
    X = region(12 + xbuf, R, C)
    y = region(i, R, C)
    dy = bco[i, 0, :C] * X + bco[i, 1, :C] + bco[i, 2, :C] * y
    ref = ys[i].grad.permute(0, 2, 3, 1).reshape(R, C)
    err = (dy - ref).abs()
    rows = (err.amax(dim=1) > 1e-4 * float(ref.abs().max())).nonzero().flatten()
    print(f"  dy{i}: err {float(err.max() / ref.abs().max()):.2e}  rows off {len(rows)} first {rows[:12].tolist()}"
          f" frames {sorted(set((rows // (P * P)).tolist()))[:10]}")


def check_w(i):
    g, t = grads[3 * i], pd[3 * i]
    print(f"  dW{i}: err {float((g - t.grad).abs().max() / t.grad.abs().max()):.2e}")


checks = {1: lambda: (check_dy(5, 0), check_dy(6, 0), check_w(5), check_w(6)),
          2: lambda: (check_dy(4, 1), check_w(4)),
          3: lambda: (check_dy(2, 2), check_dy(3, 2), check_w(2), check_w(3)),
          4: lambda: (check_dy(1, 0), check_w(1)),
          5: lambda: (check_dy(0, 1), check_w(0))}
for st in range(1, 6):
    N.check(N.lib.mivit_deepresnet_train_bwd_stage(code, ctypes.addressof(prm), x.data_ptr(), wgt.data_ptr(), n, P, E, 1e-5,
                                                   ctypes.addressof(gr), ws.data_ptr(), nbytes, st, count.data_ptr(), stats.data_ptr(), s),
            f"bwd{st}")
    torch.cuda.synchronize()
    print("after backward stage", st)
    checks[st]()
