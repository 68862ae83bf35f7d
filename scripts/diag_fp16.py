"""fp16 mode on the streaming kernels vs on the general kernels (MIVIT_NO_F16_STREAM=1) vs the reference golden values:
per-tensor gradient errors of a golden case.   python scripts/diag_fp16.py [case=c1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.nn.functional as F
from util import build_product_model, load_golden, golden_inputs
name = sys.argv[1] if len(sys.argv) > 1 else "c1"
fx, meta, cfg = load_golden(name)
params, x, labels, feats = golden_inputs(meta, cfg)
if len(sys.argv) > 2:          # a larger closed-form batch of the same shape: single ReLU flips average out
    from oracle import mivit_oracle as orc
    x, labels, feats = orc.closed_form_batch(int(sys.argv[2]), meta["T"], cfg.patch_size, cfg.global_feature_dim, salt=int(os.environ.get("DIAG_SALT", meta["salt"])))
def run(prec):
    m = build_product_model(cfg, prec, params); m.train(meta["training"]); m.zero_grad(set_to_none=True)
    out = m(x.cuda(), feats.cuda()) if feats is not None else m(x.cuda())
    loss = F.mse_loss(out, labels.cuda()); (loss * 4096.0).backward(); torch.cuda.synchronize()
    return out.detach().cpu(), float(loss), {k: (p.grad.detach() / 4096.0).cpu() for k, p in m.named_parameters()}
o32, l32, g32 = run("fp32")
o16, l16, g16 = run(os.environ.get("DIAG_PREC", "fp16"))
gs = max(float(g.abs().max()) for g in g32.values())
print("mode:", "general" if os.environ.get("MIVIT_NO_F16_STREAM") else "streaming", " loss fp32 %.6f fp16 %.6f  out err %.2e" % (l32, l16, float((o16 - o32).abs().max() / o32.abs().max())))
rows = []
for k in g32:
    d = (g16[k] - g32[k]).abs()
    rows.append((float(d.max() / (g32[k].abs().max() + 1e-3 * gs)), float(d.norm() / (g32[k].norm() + 1e-3 * gs)), k))
for mx, nr, k in sorted(rows, reverse=True)[:8]:
    print("  max-err %.3e  norm-err %.3e  %s" % (mx, nr, k))
