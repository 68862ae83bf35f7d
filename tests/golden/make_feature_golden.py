#!/usr/bin/env python3
"""Fixture for the 25 trajectory descriptors FROM THE REAL REFERENCE (build container only):

    python tests/golden/make_feature_golden.py

Imports reference helpers/helpersFeatures.py (numpy + scipy only), runs compute_diffusion_features on RNG-free
trajectories (closed-form drifting random walks of 30 and 12 points, one degenerate straight line), asserts the product's
restatement agrees, and stores inputs + the reference's outputs as numbers (tests/golden/features.npz)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from helpers import helpersFeatures as ref          # noqa: E402  (the real reference)
from moleculardiffusion_mivit_amd.helpers import features as mine    # noqa: E402


def walk(n, seed, scale, drift):
    i = np.arange(1, n + 1, dtype=np.float64)
    u = np.modf(np.sin(i * 12.9898 + seed * 78.233) * 43758.5453)[0]
    v = np.modf(np.sin(i * 39.3468 + seed * 11.135) * 24634.6345)[0]
    return np.stack([np.cumsum(u * scale) + drift * i, np.cumsum(v * scale) - 0.5 * drift * i], axis=1)


trajs = [walk(30, s, sc, dr) for s, sc, dr in [(1, 0.3, 0.0), (2, 0.05, 0.0), (3, 1.0, 0.02), (4, 0.5, 0.1), (5, 0.15, -0.03),
                                               (6, 2.0, 0.0), (7, 0.02, 0.0), (8, 0.7, 0.3)]]
trajs += [walk(12, 9, 0.4, 0.0), walk(21, 10, 0.4, 0.05)]
assert list(ref.feature_names) == mine.feature_names and ref.N_features == mine.N_features == 25
out = {}
worst = 0.0
for i, t in enumerate(trajs):
    r = np.asarray(ref.compute_diffusion_features(t, dt=1.0), dtype=np.float64)
    m = mine.compute_diffusion_features(t, dt=1.0)
    err = np.max(np.abs(r - m) / (np.abs(r) + 1e-12))
    worst = max(worst, err)
    assert err < 1e-9, (i, err, r, m)
    out[f"traj{i}"] = t
    out[f"feat{i}"] = r
avg = ref.average_trajectories_frames if hasattr(ref, "average_trajectories_frames") else None
out["n"] = np.array(len(trajs))
np.savez_compressed(os.path.join(HERE, "features.npz"), **out)
print(f"{len(trajs)} trajectories: restatement agrees with the reference to {worst:.1e}")
