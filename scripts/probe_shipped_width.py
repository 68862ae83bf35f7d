import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, torch.nn.functional as F
from oracle import mivit_oracle as orc
from util import build_product_model, rel_err
cfg = orc.MiViTConfig(embedding="linear", patch_size=13, embed_dim=64, num_heads=4, hidden_dim=128, num_layers=3)
params = orc.closed_form_params(cfg)
x, labels, _ = orc.closed_form_batch(128, 30, 13, salt=5)
res = {}
for p in ("fp32", "bf16", "fp16"):
    m = build_product_model(cfg, p, params)
    out = m(x.cuda()); loss = F.mse_loss(out, labels.cuda()); (loss * 1024.0).backward(); torch.cuda.synchronize()
    res[p] = (out.detach(), float(loss.detach()), {k: q.grad.detach() / 1024.0 for k, q in m.named_parameters()})
o32, l32, g32 = res["fp32"]
gscale = max(float(g.abs().max()) for g in g32.values())
for p in ("bf16", "fp16"):
    o, l, g = res[p]
    nr = {k: float((g[k] - g32[k]).norm() / (g32[k].norm() + 1e-3 * gscale)) for k in g32}
    w = max(nr, key=nr.get)
    print(p, "out", rel_err(o, o32), "loss", abs(l - l32) / abs(l32), "worst", w, nr[w], "median", sorted(nr.values())[len(nr)//2], "embed", {k: round(v,4) for k,v in nr.items() if k.startswith("embedding.")})
