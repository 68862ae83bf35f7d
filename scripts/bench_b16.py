"""Where a B=16 step of the reference's shipped model goes (DeepResNet embedding, 30x9x9, E64 L6, bf16): host-side split."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, torch.nn.functional as F
from moleculardiffusion_mivit_amd.helpers.models import GeneralTransformer, DeepResNetEmbedding, MLPHead
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
torch.manual_seed(0)
m = GeneralTransformer(DeepResNetEmbedding, {"patch_size": 9, "embed_dim": 64}, 64, 4, 128, 6, MLPHead, F.relu,
                       use_regression_token=True, precision=prec).cuda()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
x = torch.rand(B, 30, 9, 9, device="cuda"); y = torch.rand(B, 1, device="cuda")


def timed(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def step():
    opt.zero_grad(set_to_none=True); loss = F.mse_loss(m(x), y); loss.backward(); opt.step()
def fwd():
    with torch.no_grad(): m(x)
def fwdbwd():
    opt.zero_grad(set_to_none=True); loss = F.mse_loss(m(x), y); loss.backward()
def emb_fwdbwd():
    m.embedding.zero_grad(set_to_none=True); t = m.embedding(x); t.sum().backward()
print(f"B={B} {prec}: step {timed(step):.3f} ms | fwd+bwd {timed(fwdbwd):.3f} | train-mode fwd only (no grad) {timed(fwd):.3f} | "
      f"embedding fwd+bwd {timed(emb_fwdbwd):.3f} | optimizer {timed(lambda: opt.step()):.3f}")
