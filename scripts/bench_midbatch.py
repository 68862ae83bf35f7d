"""c1 train step at mid-size batches WITHOUT the in-library profiler (so hipGraph replay is active where it applies)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, torch.nn.functional as F
from moleculardiffusion_mivit_amd.helpers.models import GeneralTransformer, LinearProjectionEmbedding, MLPHead
torch.manual_seed(0)
m = GeneralTransformer(LinearProjectionEmbedding, {"patch_size": 64, "embed_dim": 128}, 128, 4, 256, 4, MLPHead, F.relu,
                       use_regression_token=True, precision="bf16").cuda()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
for B in [int(b) for b in (sys.argv[1:] or ["64", "256", "1024", "2048", "4096"])]:
    xs = [torch.rand(B, 32, 64, 64, device="cuda") for _ in range(2)]; y = torch.rand(B, 1, device="cuda")
    def step(i):
        opt.zero_grad(set_to_none=True); loss = F.mse_loss(m(xs[i % 2]), y); loss.backward(); opt.step()
    for i in range(8): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 40
    for i in range(n): step(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"B={B:5d}: {dt*1e3:7.3f} ms/step  {B/dt:10.0f} seq/s", flush=True)
