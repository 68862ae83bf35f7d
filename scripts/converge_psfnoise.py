"""A converged-model number: the reference's PSFNoise training loop (mirrored in experiments/PSFNoise/trainModelsPSFNoise.py:
data refresh per cycle, adaptive batch doubling every 20 cycles, AdamW(1e-4) + StepLR(5, 0.9), validation on the reference's
own val{1,3,5,7,9}.npy) run to the reference's 100 cycles for the cell its table lists first (`tr_0_0`: the MiViT with the
DeepResNet embedding, first PSF setting, no added noise), in bf16 and in the fp32 parity mode.

    python scripts/converge_psfnoise.py [cycles=100] [out=profiles/r03_converged_psfnoise.json]

Reference value beside it: outPoster/PSFNoiseResults.csv:2  tr_0_0  mse 0.2722 (std 0.129) -- its own run, its own data draw
(andi_datasets trajectories, unseeded); ours draws the same process from helpers/generation.py.  Statistical comparison, not
bitwise.  Not part of the timed benchmark."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIVIT_VALIDATION_ROOT", os.path.join(ROOT, "tests", "golden", "validation_trajectories"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from moleculardiffusion_mivit_amd.experiments.PSFNoise import trainModelsPSFNoise as loop  # noqa: E402
from moleculardiffusion_mivit_amd.experiments.PSFNoise import trainSettingsPSFNoise as S  # noqa: E402

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 100
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r03_converged_psfnoise.json")
# only the (PSF 0, noise 0) cell is trained: render only that cell (the loop renders every cell of the settings lists)
S.PSF_Settings, S.Noise_Settings = S.PSF_Settings[:1], S.Noise_Settings[:1]
S.N_PSF, S.N_Noise = 1, 1
res = {"cycles": cycles, "cell": "tr_0_0 (DeepResNet embedding, 30 x 9 x 9, E64 H4 F128 L6)", "reference_val_mse_D": 0.2722316384315491,
       "reference_source": "outPoster/PSFNoiseResults.csv:2 (100 cycles)", "validation_sets": "reference val{1,3,5,7,9}.npy (50 trajectories each)"}
for prec in ("bf16", "fp32"):
    t0 = time.time()
    models, val, labels = loop.run_training(num_cycles=cycles, N=64, seed=20251004, save=False, embedding="deepresnet", precision=prec,
                                            include_resnet=False, psf_indices=[0], noise_indices=[0])
    v = val["tr_0_0"]
    res[prec] = {"val_avg_last": v["val_avg"][-1], "val_avg_mean_last5": float(np.mean(v["val_avg"][-5:])),
                 "val_avg_first": v["val_avg"][0], "val_avg_every_10": [round(x, 4) for x in v["val_avg"][9::10]],
                 "per_set_last": {k: round(x[-1], 4) for k, x in v.items() if k != "val_avg"},
                 "sequences_seen": int(labels.shape[0]), "wall_s": round(time.time() - t0, 1)}
    print(f"[converge] {prec}: val MSE(D) {v['val_avg'][0]:.3f} -> {v['val_avg'][-1]:.4f} after {cycles} cycles "
          f"({res[prec]['wall_s']} s); reference 0.2722", flush=True)
    del models
    torch.cuda.empty_cache()
with open(out_path, "w") as fh:
    json.dump(res, fh, indent=1)
print(json.dumps(res))
