#!/usr/bin/env python3
"""Training-trajectory fixtures FROM THE REAL REFERENCE (build container only: needs /root/reference).

    python tests/golden/make_training_golden.py

Runs the reference's own modules (helpers/models.py GeneralTransformer + nn.MSELoss + optim.AdamW(lr=1e-4) +
StepLR(5, 0.9), i.e. the step sequence of Experiments/PSFNoise/trainModelsPSFNoise.py:187-196 and the evaluation of
:224-229) for a few dozen optimizer steps on RNG-free inputs, and stores ONLY numbers: the loss of every step, the
learning rate schedule, the validation MSE(D) (pred * 10 vs D) before and after, and parameter norms / samples at the
end.  Inputs and initial weights are regenerated on the GPU box by oracle/mivit_oracle.py (closed_form_params,
closed_form_batch, synthetic_batch), so nothing of the reference travels.  The oracle is run alongside and must agree.

Cases:
  train_c1                 BASELINE configs[0/1] shape (32 x 64 x 64, depth 4, dim 128, linear embedding), B = 8, 50 steps
  train_psfnoise_drn       reduced PSFNoise model (DeepResNet embedding, 9 x 9 frames, E 64, depth 2), B = 8, T = 10, 40 steps,
                           O(1) normalised frames
  train_psfnoise_drn_counts   the same on camera-count scale frames (background ~5000 + spot ~5000: what the reference's
                           PSFNoise loop actually feeds, no normalize_images call)
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

from oracle import mivit_oracle as orc            # noqa: E402
from make_golden import build_reference, sample_idx  # noqa: E402  (imports the real reference)

torch.set_num_threads(8)

CASES = {
    "train_c1": dict(cfg=dict(embedding="linear", patch_size=64, embed_dim=128, num_heads=4, hidden_dim=256, num_layers=4),
                     B=8, T=32, steps=50, nbatches=4, data="closed_form", scale=None),
    "train_psfnoise_drn": dict(cfg=dict(embedding="deepresnet", patch_size=9, embed_dim=64, num_heads=4, hidden_dim=128,
                                        num_layers=2), B=8, T=10, steps=40, nbatches=5, data="synthetic", scale=None),
    "train_psfnoise_drn_counts": dict(cfg=dict(embedding="deepresnet", patch_size=9, embed_dim=64, num_heads=4, hidden_dim=128,
                                               num_layers=2), B=8, T=10, steps=40, nbatches=5, data="synthetic",
                                      scale=(5000.0, 8000.0)),
}
STEPS_PER_CYCLE = 10        # StepLR.step() once per "cycle" of this many optimizer steps (trainModelsPSFNoise.py:199)


def batches(spec, cfg):
    """(train batches, validation batch) -- the same generator the GPU test calls."""
    out = []
    for i in range(spec["nbatches"] + 1):
        if spec["data"] == "closed_form":
            x, y, _ = orc.closed_form_batch(spec["B"], spec["T"], cfg.patch_size, salt=i)
        else:
            x, y, _ = orc.synthetic_batch(spec["B"], spec["T"], cfg.patch_size, seed=4200 + i)
        if spec["scale"]:
            x = spec["scale"][0] + spec["scale"][1] * x
        out.append((x, y))
    return out[:-1], out[-1]


def run(name, spec):
    cfg = orc.MiViTConfig(**spec["cfg"])
    params = orc.closed_form_params(cfg)
    train, (xv, yv) = batches(spec, cfg)
    m = build_reference(cfg)
    sd = m.state_dict()
    m.load_state_dict({**{k: v for k, v in sd.items() if k.endswith("num_batches_tracked")}, **params})
    om = orc.OracleModule(cfg, params)
    res = {}
    for tag, model in (("ref", m), ("oracle", om)):
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
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
        losses, lrs = [], []
        for s in range(spec["steps"]):
            x, y = train[s % len(train)]
            opt.zero_grad()
            loss = lossf(model(x), y)
            loss.backward()
            opt.step()
            losses.append(float(loss))
            lrs.append(opt.param_groups[0]["lr"])
            if (s + 1) % STEPS_PER_CYCLE == 0:
                sch.step()
        res[tag] = dict(losses=np.array(losses), lrs=np.array(lrs), val0=v0, val1=val())
    r, o = res["ref"], res["oracle"]
    e_traj = float(np.max(np.abs(r["losses"] - o["losses"]) / np.abs(r["losses"])))
    e_val = abs(r["val1"] - o["val1"]) / abs(r["val1"])
    print(f"{name:28s} loss {r['losses'][0]:.6f} -> {r['losses'][-1]:.6f}   val MSE(D) {r['val0']:.5f} -> {r['val1']:.5f}   "
          f"oracle-vs-reference: trajectory {e_traj:.1e}, val {e_val:.1e}")
    # AdamW's first updates are ~ lr * sign(g): two correct fp32 implementations that differ by rounding in a near-zero
    # gradient take different steps there, and the trajectories drift apart by far more than 1e-4 within tens of steps.
    # The fixture records that floor (two CPU fp32 runs of the SAME arithmetic: the reference modules vs the functional
    # restatement); the GPU tests hold the early steps to 1e-4 and the rest to a band tied to it.
    e_first = float(np.max(np.abs(r["losses"][:3] - o["losses"][:3]) / np.abs(r["losses"][:3])))
    print(f"{'':28s} first 3 steps {e_first:.1e};  per-decade divergence " +
          " ".join(f"{float(np.max(np.abs(r['losses'][i:i + 10] - o['losses'][i:i + 10]) / np.abs(r['losses'][i:i + 10]))):.1e}"
                   for i in range(0, spec["steps"], 10)))
    assert e_first < 2e-3 and e_traj < 5e-2 and e_val < 5e-2, (name, e_first, e_traj, e_val)
    fx = {"losses": r["losses"], "lrs": r["lrs"], "val_mse_D_before": np.float64(r["val0"]), "val_mse_D_after": np.float64(r["val1"])}
    for k, p in m.named_parameters():
        flat = p.detach().reshape(-1).numpy()
        fx["pnorm/" + k] = np.float64(np.linalg.norm(flat.astype(np.float64)))
        fx["psamp/" + k] = flat[sample_idx(flat.size)]
    meta = {"config": cfg.to_dict(), "B": spec["B"], "T": spec["T"], "steps": spec["steps"], "nbatches": spec["nbatches"],
            "data": spec["data"], "scale": spec["scale"], "steps_per_cycle": STEPS_PER_CYCLE, "lr": 1e-4, "torch": torch.__version__,
            "cpu_fp32_floor": {"trajectory": e_traj, "val": e_val, "first3": e_first}}
    fx["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)


if __name__ == "__main__":
    only = sys.argv[1:]
    for nm, spec in CASES.items():
        if not only or nm in only:
            run(nm, spec)
