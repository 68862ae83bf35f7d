"""Diagnostic: worst gradient error of the native DeepResNet training step vs the fp64 torch stack over frame counts."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculardiffusion_mivit_amd.helpers.models import DeepResNetEmbedding

P = int(sys.argv[1]); prec = sys.argv[2]
for n in [int(v) for v in sys.argv[3].split(",")]:
    torch.manual_seed(7)
    nat = DeepResNetEmbedding(P, 64).cuda().train()
    nat.__dict__["_mivit_precision"] = prec
    t64 = copy.deepcopy(nat).double()
    x = torch.rand(1, n, P, P, device="cuda") * 1.5 - 0.25
    wgt = torch.randn(1, n, 64, device="cuda")
    o_n = nat(x); (o_n * wgt).sum().backward()
    os.environ["MIVIT_NO_DEEPRESNET_TRAIN"] = "1"
    o_64 = t64(x.double()); (o_64 * wgt.double()).sum().backward()
    os.environ.pop("MIVIT_NO_DEEPRESNET_TRAIN")
    ref = {k: p.grad for k, p in t64.named_parameters()}
    gscale = max(float(g.abs().max()) for g in ref.values())
    errs = {k: float((p.grad - ref[k]).abs().max()) / (float(ref[k].abs().max()) + 1e-3 * gscale) for k, p in nat.named_parameters()}
    kmax = max(errs, key=errs.get)
    print(f"P {P} N {n:5d} tokens {float((o_n.detach() - o_64.detach()).abs().max() / o_64.detach().abs().max()):.1e} "
          f"bn2.bias {errs['res_block2.bn2.bias']:.1e} worst {errs[kmax]:.1e} ({kmax})", flush=True)
