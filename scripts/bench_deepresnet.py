"""Training step of the reference's shipped model (DeepResNet embedding, 30 frames of 9x9, E64 H4 F128 L6):
hand-written conv / BatchNorm kernels vs the PyTorch-ROCm (MIOpen) conv stack, same module, same optimiser."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, torch.nn.functional as F
from moleculardiffusion_mivit_amd.helpers.models import GeneralTransformer, DeepResNetEmbedding, MLPHead

def make(prec):
    torch.manual_seed(0)
    return GeneralTransformer(DeepResNetEmbedding, {"patch_size": 9, "embed_dim": 64}, 64, 4, 128, 6, MLPHead, F.relu,
                              use_regression_token=True, precision=prec).cuda()

batches = [int(b) for b in os.environ.get("BATCHES", "16,256,1024").split(",")]
for prec in os.environ.get("PRECS", "fp32,bf16").split(","):
    for B in batches:
        x = torch.rand(B, 30, 9, 9, device="cuda"); y = torch.rand(B, 1, device="cuda")
        res = {}
        for native in ((True,) if os.environ.get('NATIVE_ONLY') else (True, False)):
            os.environ.pop("MIVIT_NO_DEEPRESNET_TRAIN", None)
            if not native:
                os.environ["MIVIT_NO_DEEPRESNET_TRAIN"] = "1"
            m = make(prec); opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
            def step():
                opt.zero_grad(set_to_none=True); loss = F.mse_loss(m(x), y); loss.backward(); opt.step()
            for _ in range(3): step()
            n = 20 if B <= 256 else 5
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): step()
            torch.cuda.synchronize(); res[native] = (time.perf_counter() - t0) / n
        res.setdefault(False, float("nan"))
        print(f"{prec} B={B:5d}: native {res[True]*1e3:8.2f} ms ({B/res[True]:9.0f} seq/s)   MIOpen stack {res[False]*1e3:8.2f} ms "
              f"({B/res[False]:9.0f} seq/s)   speed-up {res[False]/res[True]:.2f}x", flush=True)
