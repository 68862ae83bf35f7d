"""Train-step time and model TFLOP/s of the BASELINE configurations other than the headline one (1 GPU):
c4 = Embeddings sweep (T=64, E=512, H=8, F=1024, L=4, P in {16, 64, 128}), c3 = Framerate shape (P=13, T=30/60),
c5 = ImagesFeatures (c1 shape + 25 features, early / late fusion, fp16 + GradScaler and bf16)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, torch.nn.functional as F
from moleculardiffusion_mivit_amd import _native as N
from moleculardiffusion_mivit_amd.helpers.models import GeneralTransformer, LinearProjectionEmbedding, MLPHead
import ctypes

def flops_train(T, P, E, H, Fh, L, reg=True):
    S = T + (1 if reg else 0)
    fwd = 2 * T * P * P * E + L * (6 * S * E * E + 4 * S * S * E + 2 * S * E * E + 4 * S * E * Fh) + 2 * E * 128 + 256
    return 3 * fwd

def run(name, B, T, P, E, H, Fh, L, pos=False, prec="bf16", steps=8, fusion=None):
    torch.manual_seed(0)
    kw = dict(use_global_features=True, fusion_type=fusion, global_feature_dim=25) if fusion else {}
    m = GeneralTransformer(LinearProjectionEmbedding, {"patch_size": P, "embed_dim": E}, E, H, Fh, L, MLPHead, F.relu,
                           use_regression_token=True, use_pos_encoding=pos, precision=prec, **kw).cuda()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
    x = torch.rand(B, T, P, P, device="cuda"); y = torch.rand(B, 1, device="cuda")
    feats = torch.randn(B, 25, device="cuda") if fusion else None
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 16, growth_interval=2000) if prec == "fp16" else None   # config 5: fp16 + loss scaling
    def step():
        opt.zero_grad(set_to_none=True); loss = F.mse_loss(m(x, feats) if fusion else m(x), y)
        if scaler is None:
            loss.backward(); opt.step()
        else:
            scaler.scale(loss).backward(); scaler.step(opt); scaler.update()
    for _ in range(3): step()
    torch.cuda.synchronize()
    tags = (1 << len(N.PROF_TAGS)) - 1
    N.lib.mivit_profile_enable(ctypes.c_uint64(tags))
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    cat = {}
    for tag in range(len(N.PROF_TAGS)):
        ms, n = ctypes.c_double(), ctypes.c_int()
        N.lib.mivit_profile_collect(tag, ctypes.byref(ms), ctypes.byref(n))
        nm = N.lib.mivit_profile_tag_name(tag)
        cat[nm.decode() if nm else str(tag)] = round(ms.value / steps, 3)
    N.lib.mivit_profile_enable(ctypes.c_uint64(0))
    fl = flops_train(T, P, E, H, Fh, L) * B
    cat = {k: v for k, v in cat.items() if v > 0}
    print(f"{name:34s} B={B:5d}: {dt*1e3:8.2f} ms/step  {B/dt:10.0f} seq/s  {fl/dt/1e12:7.1f} model-TFLOP/s   {cat}", flush=True)

if __name__ == "__main__":
    which = sys.argv[1:] or ["c4_16", "c4_64", "c4_128", "c3", "c5"]
    if "c4_16" in which: run("c4 P=16 T=64 E=512 L=4", 2048, 64, 16, 512, 8, 1024, 4, pos=True)
    if "c4_64" in which: run("c4 P=64 T=64 E=512 L=4", 1024, 64, 64, 512, 8, 1024, 4, pos=True)
    if "c4_128" in which: run("c4 P=128 T=64 E=512 L=4", 256, 64, 128, 512, 8, 1024, 4, pos=True)
    if "ref" in which:
        for B in (16, 64, 256, 1024):
            run("ref shape P=9 T=30 E=64 L=6", B, 30, 9, 64, 4, 128, 6, steps=30)
    if "c5" in which:       # ImagesFeatures: c1 shape + 25 trajectory features, early / late fusion; fp16 with loss scaling as BASELINE names it, bf16 beside it
        for prec in ("fp16", "bf16"):
            for fusion in ("early", "late"):
                run(f"c5 {fusion} fusion {prec} P=64 T=32 E=128", 4096, 32, 64, 128, 4, 256, 4, prec=prec, fusion=fusion)
    if "c3" in which:
        run("c3 Framerate P=13 T=30 E=64", 4096, 30, 13, 64, 4, 128, 6)
        run("c3 Framerate P=13 T=60 E=64", 4096, 60, 13, 64, 4, 128, 6)
