"""Config 3 (Framerate shape: P=13, T=30, E=64, H=4, F=128, L=6) train-step time without the in-library profiler (hipGraph
replay stays on).  python scripts/bench_c3.py [B=4096] [T=30] [steps=20]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, torch.nn.functional as F
from moleculardiffusion_mivit_amd.helpers.models import GeneralTransformer, LinearProjectionEmbedding, MLPHead
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
torch.manual_seed(0)
m = GeneralTransformer(LinearProjectionEmbedding, {"patch_size": 13, "embed_dim": 64}, 64, 4, 128, 6, MLPHead, F.relu,
                       use_regression_token=True, precision="bf16").cuda()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
x, y = torch.rand(B, T, 13, 13, device="cuda"), torch.rand(B, 1, device="cuda")
def step():
    opt.zero_grad(set_to_none=True); F.mse_loss(m(x), y).backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"c3 P=13 T={T} E=64 L=6 B={B}: {dt * 1e3:.3f} ms/step, {B / dt:.0f} seq/s (MIVIT_GRAPHS={os.environ.get('MIVIT_GRAPHS', '1')})")
