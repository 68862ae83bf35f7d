"""Training trajectories of the HIP path against fixtures made by the REAL reference (tests/golden/make_training_golden.py:
reference GeneralTransformer + nn.MSELoss + optim.AdamW(lr=1e-4) + StepLR(5, 0.9); reference
Experiments/PSFNoise/trainModelsPSFNoise.py:187-196 step sequence, :224-229 validation MSE(D) = MSE(pred * 10, D)).

Tolerances (relative, on the loss of every step and on the final validation MSE(D)):
  fp32 mode : the first 3 steps within max(1e-4, 3 x floor); every step within max(2e-3, 4 x floor); val within max(5e-3, 4 x floor)
              -- "floor" is the divergence the fixture itself records between two CPU fp32 runs of the same arithmetic
              (reference modules vs the functional oracle): AdamW's sign-like first updates amplify rounding noise in near-zero
              gradients, so 1e-4 does not survive tens of steps even between two CPU implementations (and the PyTorch-ROCm /
              MIOpen stack run through the same schedule sits 1.1e-3 .. 1.3e-3 from the reference on the DeepResNet case,
              differently on every run; this path: 1.16e-3, bitwise repeatable);
  bf16 mode : mean step error within 6e-2; final validation MSE(D) within 4e-2 for the transformer-only model (north_star:
              val-loss parity; measured 1.6-2.7e-2, PyTorch's bf16 autocast 2.8e-2) and within 8e-2 with the BatchNorm'd conv stack in front (80 frames per step: every stored
              activation carries 2^-9 relative error and AdamW amplifies it).  PyTorch's own bf16 autocast of the SAME
              arithmetic (the oracle module on the GPU) is run beside it and printed as a yardstick.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import mivit_oracle as orc
from util import GOLDEN, build_product_model

pytestmark = pytest.mark.gpu

CASES = ["train_c1", "train_psfnoise_drn", "train_psfnoise_drn_counts"]


def _batches(meta, cfg):
    out = []
    for i in range(meta["nbatches"] + 1):
        if meta["data"] == "closed_form":
            x, y, _ = orc.closed_form_batch(meta["B"], meta["T"], cfg.patch_size, salt=i)
        else:
            x, y, _ = orc.synthetic_batch(meta["B"], meta["T"], cfg.patch_size, seed=4200 + i)
        if meta["scale"]:
            x = meta["scale"][0] + meta["scale"][1] * x
        out.append((x.cuda(), y.cuda()))
    return out[:-1], out[-1]


def run_schedule(model, meta, cfg):
    train, (xv, yv) = _batches(meta, cfg)
    opt = torch.optim.AdamW(model.parameters(), lr=meta["lr"])
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=5, gamma=0.9)
    lossf = nn.MSELoss()

    def val():
        model.eval()
        with torch.no_grad():
            v = float(lossf(model(xv) * 10.0, yv * 10.0))
        model.train()
        return v
    model.train()
    v0 = val()
    losses = []
    for s in range(meta["steps"]):
        x, y = train[s % len(train)]
        opt.zero_grad()
        loss = lossf(model(x), y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        if (s + 1) % meta["steps_per_cycle"] == 0:
            sch.step()
    return np.array(losses), v0, val()


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_training_trajectory_matches_reference(name, precision):
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(fx["meta"]))
    cfg = orc.MiViTConfig(**meta["config"])
    model = build_product_model(cfg, precision, orc.closed_form_params(cfg), device="cuda")
    losses, v0, v1 = run_schedule(model, meta, cfg)
    ref = fx["losses"]
    err = np.abs(losses - ref) / np.abs(ref)
    e_v0 = abs(v0 - float(fx["val_mse_D_before"])) / float(fx["val_mse_D_before"])
    e_v1 = abs(v1 - float(fx["val_mse_D_after"])) / float(fx["val_mse_D_after"])
    print(f"{name} {precision}: step err first3 {err[:3].max():.2e} all {err.max():.2e}; val before {e_v0:.2e} after {e_v1:.2e} "
          f"(val MSE(D) {v1:.4f} vs reference {float(fx['val_mse_D_after']):.4f})")
    assert losses[-1] < 0.8 * losses[0]                      # it learns
    if precision == "fp32":
        floor = meta["cpu_fp32_floor"]
        assert e_v0 < 1e-4
        assert err[:3].max() < max(1e-4, 3 * floor["first3"])
        assert err.max() < max(2e-3, 4 * floor["trajectory"])
        assert e_v1 < max(5e-3, 4 * floor["val"])
    else:
        # yardstick: plain PyTorch bf16 autocast of the same arithmetic through the same schedule
        om = orc.OracleModule(cfg, orc.closed_form_params(cfg)).cuda()

        class Autocast(torch.nn.Module):
            def __init__(self, inner):
                super().__init__()
                self.inner = inner

            def forward(self, x):
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    return self.inner(x).float()
        yl, _, yv1 = run_schedule(Autocast(om), meta, cfg)
        y_all = np.abs(yl - ref) / np.abs(ref)
        y_val = abs(yv1 - float(fx["val_mse_D_after"])) / float(fx["val_mse_D_after"])
        print(f"    torch bf16 autocast of the oracle: step err max {y_all.max():.2e} mean {y_all.mean():.2e}, val {y_val:.2e};  "
              f"this path: max {err.max():.2e} mean {err.mean():.2e}")
        assert e_v0 < 2e-2
        # max over steps is dominated by AdamW's first, sign-like updates (the reference's own loss jumps 0.12 -> 1.29 -> 0.10
        # on the c1 schedule): the band is on the MEAN step error, the early transient included.  The autocast yardstick is
        # printed, not asserted against: through MIOpen's non-deterministic kernels it lands anywhere between 1 % and 13 % on
        # the DeepResNet cases' validation loss from run to run (2.8 % on c1); this path is bitwise repeatable (measured: c1 mean
        # 4.4e-2 / val 1.6e-2 .. 2.7e-2 across two generations of the backward kernels -- the schedule is chaotic at that level; DeepResNet mean 3.9e-2 / val 6.1e-2; DeepResNet on camera counts mean 1.2e-2 / val 2.3e-2).
        assert err.mean() < 6e-2
        assert e_v1 < (4e-2 if cfg.embedding != "deepresnet" else 8e-2)
