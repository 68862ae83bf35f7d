"""Kernel launches of one reference-shaped train step (B=16, 30x9x9, E64 L6, linear embedding, bf16) for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, torch.nn.functional as F
from moleculardiffusion_mivit_amd.helpers.models import GeneralTransformer, LinearProjectionEmbedding, MLPHead
os.environ.setdefault("MIVIT_GRAPHS", "0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
m = GeneralTransformer(LinearProjectionEmbedding, {"patch_size": 9, "embed_dim": 64}, 64, 4, 128, 6, MLPHead, F.relu,
                       use_regression_token=True, precision="bf16").cuda()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
x = torch.rand(B, 30, 9, 9, device="cuda"); y = torch.rand(B, 1, device="cuda")
for _ in range(10):
    opt.zero_grad(set_to_none=True); loss = F.mse_loss(m(x), y); loss.backward(); opt.step()
torch.cuda.synchronize()
