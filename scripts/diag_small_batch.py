"""Where a launch-bound train step goes (config 3 at the reference's own batch sizes): wall time per phase with a device sync
after each, the un-synchronised step beside it, and the host-side profile of the step.  python scripts/diag_small_batch.py [B=32]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, torch.nn.functional as F
from moleculardiffusion_mivit_amd.helpers.models import GeneralTransformer, LinearProjectionEmbedding, MLPHead
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = GeneralTransformer(LinearProjectionEmbedding, {"patch_size": 13, "embed_dim": 64}, 64, 4, 128, 6, MLPHead, F.relu,
                       use_regression_token=True, precision="bf16").cuda()
if os.environ.get("DIRECT", "0") == "1":
    m.direct_param_grads(True)
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
x, y = torch.rand(B, 30, 13, 13, device="cuda"), torch.rand(B, 1, device="cuda")
sync = torch.cuda.synchronize
def step():
    opt.zero_grad(set_to_none=True); F.mse_loss(m(x), y).backward(); opt.step()
for _ in range(20): step()
sync()
n = 200
t0 = time.perf_counter()
for _ in range(n): step()
sync(); total = (time.perf_counter() - t0) / n
seg = {"zero_grad": 0.0, "forward": 0.0, "loss": 0.0, "backward": 0.0, "opt.step": 0.0}
host = dict(seg)
for _ in range(n):
    for name, fn in (("zero_grad", lambda: opt.zero_grad(set_to_none=True)), ("forward", None), ("loss", None), ("backward", None), ("opt.step", opt.step)):
        t = time.perf_counter()
        if name == "forward": out = m(x)
        elif name == "loss": loss = F.mse_loss(out, y)
        elif name == "backward": loss.backward()
        else: fn()
        h = time.perf_counter(); sync(); e = time.perf_counter()
        host[name] += h - t; seg[name] += e - t
print(f"B={B}: un-synchronised step {total * 1e3:.3f} ms; params {sum(1 for _ in m.parameters())} tensors")
for k in seg:
    print(f"  {k:10s} host {host[k] / n * 1e3:7.3f} ms   host+device {seg[k] / n * 1e3:7.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
sync(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
