"""Real-data patch extraction (reference helpers/helpersTracking.py:513-550 ``extract_particle_patches``): square patches around
every tracked position, zero-padded at the image border -- the tensors a trained MiViT consumes on experimental movies
(SURVEY section 8 row f4).  One gather for all positions of a track; works on CPU or GPU tensors."""
from typing import Dict, Sequence, Tuple

import numpy as np
import torch


def extract_particle_patches(image_3d, tracks: Dict[object, Sequence[Tuple[int, float, float]]], patch_size: int = 7):
    """image_3d (num_frames, H, W) array / tensor; tracks {id: [(frame, y, x), ...]} -> {id: (len, patch, patch)} of the input's
    kind (numpy in, numpy out).  Positions are rounded to the nearest pixel; pixels outside the image read as 0."""
    assert patch_size % 2 == 1, "patch_size must be an odd number"
    is_np = not torch.is_tensor(image_3d)
    img = torch.as_tensor(np.asarray(image_3d)) if is_np else image_3d
    half = patch_size // 2
    nf, H, W = img.shape
    off = torch.arange(-half, half + 1, device=img.device)
    out = {}
    for tid, positions in tracks.items():
        if len(positions) == 0:
            out[tid] = np.array([]) if is_np else img.new_zeros((0, patch_size, patch_size))
            continue
        pos = np.asarray(positions, dtype=np.float64)
        fr = torch.as_tensor(pos[:, 0].astype(np.int64), device=img.device)
        # Python's round() (round-half-to-even), like the reference's int(round(y))
        yy = torch.as_tensor(np.rint(pos[:, 1]).astype(np.int64), device=img.device)
        xx = torch.as_tensor(np.rint(pos[:, 2]).astype(np.int64), device=img.device)
        ys = yy[:, None] + off[None, :]                       # (L, p)
        xs = xx[:, None] + off[None, :]
        ok = ((ys >= 0) & (ys < H))[:, :, None] & ((xs >= 0) & (xs < W))[:, None, :]
        g = img[fr[:, None, None], ys.clamp(0, H - 1)[:, :, None], xs.clamp(0, W - 1)[:, None, :]]
        g = torch.where(ok, g, torch.zeros((), dtype=img.dtype, device=img.device))
        out[tid] = g.numpy() if is_np else g
    return out
