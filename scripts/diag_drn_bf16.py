"""bf16 error of the native DeepResNet training kernels vs torch bf16 autocast, both against the fp32 torch stack."""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculardiffusion_mivit_amd.helpers.models import DeepResNetEmbedding

def run(P, B, T, E):
    torch.manual_seed(P * 1000 + B * 10 + T)
    ref = DeepResNetEmbedding(P, E).cuda().train()
    nat = copy.deepcopy(ref); nat.__dict__["_mivit_precision"] = "bf16"
    ac = copy.deepcopy(ref)
    x = torch.rand(B, T, P, P, device="cuda") * 1.5 - 0.25
    wgt = torch.randn(B, T, E, device="cuda")
    os.environ["MIVIT_NO_DEEPRESNET_TRAIN"] = "1"
    o_ref = ref(x); (o_ref * wgt).sum().backward()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o_ac = ac(x)
    (o_ac.float() * wgt).sum().backward()
    del os.environ["MIVIT_NO_DEEPRESNET_TRAIN"]
    o_nat = nat(x); (o_nat * wgt).sum().backward()
    torch.cuda.synchronize()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    print(f"P={P} B={B} T={T} E={E}: out native {rel(o_nat, o_ref):.3e} autocast {rel(o_ac.float(), o_ref):.3e}")
    for (k, pr), (_, pn), (_, pa) in zip(ref.named_parameters(), nat.named_parameters(), ac.named_parameters()):
        print(f"   {k:28s} native {rel(pn.grad, pr.grad):.3e}  autocast {rel(pa.grad, pr.grad):.3e}  |g| {float(pr.grad.abs().max()):.3e}")

run(9, 4, 30, 64)
run(13, 2, 3, 128)
