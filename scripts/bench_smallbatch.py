"""Latency of the reference-shaped training step (B=16, T=30, 9x9 frames, E=64 H=4 F=128 L=6, linear embedding) --
the regime the reference's own loops run in (trainModelsPSFNoise.py: batch 1..16) -- and of in-order inference."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, torch.nn.functional as F
from moleculardiffusion_mivit_amd.helpers.models import GeneralTransformer, LinearProjectionEmbedding, DeepResNetEmbedding, MLPHead

def make(emb, prec):
    torch.manual_seed(0)
    return GeneralTransformer(emb, {"patch_size": 9, "embed_dim": 64}, 64, 4, 128, 6, MLPHead, F.relu,
                              use_regression_token=True, precision=prec).cuda()

for emb in (LinearProjectionEmbedding, DeepResNetEmbedding):
    for prec in ("fp32", "bf16"):
        m = make(emb, prec); opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
        for B in (1, 16):
            x = torch.rand(B, 30, 9, 9, device="cuda"); y = torch.rand(B, 1, device="cuda")
            def step():
                opt.zero_grad(set_to_none=True); loss = F.mse_loss(m(x), y); loss.backward(); opt.step()
            for _ in range(5): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): step()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
            print(f"train step {emb.__name__:26s} {prec} B={B:2d}: {dt*1e3:7.3f} ms")
        m.eval()
        xv = torch.rand(1000, 30, 9, 9, device="cuda")
        with torch.no_grad():
            for _ in range(3): m(xv)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): m(xv)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"inference  {emb.__name__:26s} {prec} 1000 sequences of 30x9x9: {dt*1e3:7.2f} ms   (reference notebook: 9200 ms for im_tr on its in-order set, hardware unspecified)")
