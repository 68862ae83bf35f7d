"""Synthetic single-molecule data for the training loops (the step right before the hot path).

Own, vectorised, seedable restatement of what the reference's generator produces, written with torch ops so it runs
on the host or directly on the GPU (no andi_datasets / skimage dependency):

* ``brownian_single_state``  stands in for ``andi_datasets.models_phenom().single_state(N, L=0, T, Ds=[mean, var],
  alphas=1)`` as consumed at ``Experiments/PSFNoise/trainModelsPSFNoise.py:128-142``: free Brownian motion,
  increments ``sqrt(2 D dt) N(0,1)`` (the law spelled out in ``mitochondria_simulation/mitochnodria.py:470-474``),
  per-particle ``D ~ N(mean, var)`` redrawn until positive; returns ``(T, N, 2)`` trajectories and ``(T, N, 3)``
  labels ``[alpha, D, state]``.
* ``render_frames`` is the noiseless image model of ``helpers/helpersGeneration.py:283-308`` /
  ``trainSettingsPSFNoise.py:265-292``: each frame is the sum of ``nPosPerFrame`` peak-normalised Gaussian spots on a
  ``upsampling_factor``-times finer grid, mean-pooled back.  A peak-normalised 2-D Gaussian on a grid is an outer
  product of two 1-D profiles, and mean-pooling an outer product is the outer product of the pooled profiles, so a
  frame costs O(p * (G + P^2)) instead of O(p * G^2) and the Python triple loop disappears.
* noise models: clipped-Gaussian background + Poisson, in both variants the reference uses.
* ``normalize_images`` = ``helpersGeneration.py:356-400``.

The reference draws from the unseeded global numpy RNG, so agreement is distributional; the deterministic part
(noise-free rendering) is pinned against a naive per-pixel loop in ``tests/test_generation.py``.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch

DEFAULT_IMAGE_PROPS = {
    "particle_intensity": [500, 20],
    "NA": 1.46,
    "wavelength": 500e-9,
    "psf_division_factor": 1,
    "resolution": 100e-9,
    "output_size": 32,
    "upsampling_factor": 5,
    "background_intensity": [100, 10],
    "poisson_noise": 100,
    "trajectory_unit": 100,
}


def _as_tensor(x, device=None, dtype=torch.float32):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype)
    return torch.as_tensor(x, dtype=dtype, device=device)


def brownian_single_state(N: int, T: int, Ds=(1.0, 0.0), alphas: float = 1.0, dt: float = 1.0,
                          generator: Optional[torch.Generator] = None, device="cpu"):
    """(T, N, 2) trajectories and (T, N, 3) labels [alpha, D, state=0] of freely diffusing particles."""
    mean, var = float(Ds[0]), float(Ds[1])
    D = torch.full((N,), mean, device=device)
    if var > 0:
        D = mean + math.sqrt(var) * torch.randn(N, generator=generator, device=device)
        for _ in range(64):                      # redraw non-positive coefficients (AnDi constrains D > 0)
            bad = D <= 1e-4
            if not bool(bad.any()):
                break
            D = torch.where(bad, mean + math.sqrt(var) * torch.randn(N, generator=generator, device=device), D)
        D = D.clamp_min(1e-4)
    steps = torch.randn(T, N, 2, generator=generator, device=device) * torch.sqrt(2.0 * D * dt).view(1, N, 1)
    steps[0] = 0.0
    trajs = torch.cumsum(steps, dim=0)
    labels = torch.stack([torch.full((T, N), float(alphas), device=device), D.view(1, N).expand(T, N),
                          torch.zeros(T, N, device=device)], dim=-1)
    return trajs, labels


def psf_sigma_hr(props: dict) -> float:
    """Gaussian sigma on the upsampled grid (helpersGeneration.py: fwhm = wavelength / 2 * NA / psf_division_factor)."""
    fwhm = props["wavelength"] / 2 * props["NA"] / props.get("psf_division_factor", 1)
    return props["upsampling_factor"] / props["resolution"] * fwhm / 2.355


def render_frames(traj_px: torch.Tensor, nPosPerFrame: int, sigmas: Sequence[float], output_size: int,
                  upsampling_factor: int, spot_intensity: torch.Tensor, center: bool = False) -> torch.Tensor:
    """Noise-free frames.  traj_px: (N, T, 2) in pixels; sigmas: one hi-res sigma per PSF setting;
    spot_intensity: (N, F, p) amplitude of every sub-position.  Returns (N, len(sigmas), F, P, P)."""
    N, T, _ = traj_px.shape
    if T % nPosPerFrame != 0:
        raise Exception("T is not divisble by posPerFrame")
    if traj_px.device.type == "cuda":
        return _render_frames_hip(traj_px, nPosPerFrame, sigmas, output_size, upsampling_factor, spot_intensity, center)
    F_, p, P, up = T // nPosPerFrame, nPosPerFrame, output_size, upsampling_factor
    G = P * up
    dev = traj_px.device
    seg = traj_px.reshape(N, F_, p, 2)
    if center:
        seg = seg - seg.mean(dim=2, keepdim=True)
    seg = seg * up
    limit = (G - 1) // 2                                   # same (integer) limit as the reference grid
    axis = torch.linspace(-limit, limit, G, device=dev, dtype=traj_px.dtype)
    out = []
    for s in sigmas:
        # spot / spot_max per axis, as ONE exponential of the difference of squared distances: the reference divides two
        # float64 Gaussians (a spot that left the frame still peaks at full intensity on the border); fp32 factors would
        # underflow to 0 / 0 there
        dx2 = (axis.view(1, 1, 1, G) - seg[..., 0:1]) ** 2
        dy2 = (axis.view(1, 1, 1, G) - seg[..., 1:2]) ** 2
        prof = torch.exp(-(dx2 - dx2.amin(dim=-1, keepdim=True)) / (2 * s * s))       # x profile (N,F,p,G), peak-normalised
        profy = torch.exp(-(dy2 - dy2.amin(dim=-1, keepdim=True)) / (2 * s * s))      # y profile
        px = prof.reshape(N, F_, p, P, up).mean(dim=-1)                   # mean pooling of each 1-D profile
        py = profy.reshape(N, F_, p, P, up).mean(dim=-1)
        # frame[y, x] = sum_p a_p * py_p[y] * px_p[x]
        out.append(torch.einsum("nfp,nfpy,nfpx->nfyx", spot_intensity, py, px))
    return torch.stack(out, dim=1)


def _render_frames_hip(traj_px, nPosPerFrame, sigmas, output_size, upsampling_factor, spot_intensity, center):
    """GPU tensors: the hand-written kernel (csrc/render.hip, mivit_render_frames) -- one workgroup per (sequence, frame, PSF)."""
    import ctypes
    from .. import _native as Nat
    N, T, _ = traj_px.shape
    F_ = T // nPosPerFrame
    dev = traj_px.device
    traj = traj_px.contiguous().float()
    amp = spot_intensity.to(dev).expand(N, F_, nPosPerFrame).contiguous().float()
    sig = torch.tensor([float(s) for s in sigmas], dtype=torch.float32, device=dev)
    out = torch.empty(N, len(sig), F_, output_size, output_size, dtype=torch.float32, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    for n0 in range(0, N, 65535):                     # grid.y limit
        n1 = min(N, n0 + 65535)
        Nat.check(Nat.lib.mivit_render_frames(vp(traj[n0:n1]), n1 - n0, T, nPosPerFrame, vp(sig), len(sig), output_size,
                                              upsampling_factor, vp(amp[n0:n1]), int(bool(center)), vp(out[n0:n1]),
                                              ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mivit_render_frames")
    return out.to(traj_px.dtype)


def clipped_background(shape, mean: float, std: float, generator=None, device="cpu"):
    """np.clip(np.random.normal(mean, std), 0, mean + 3 std)  (helpersGeneration.py:312-313)."""
    if std <= 0:
        return torch.full(shape, float(min(max(mean, 0.0), mean)), device=device)
    return (mean + std * torch.randn(shape, generator=generator, device=device)).clamp(0.0, mean + 3 * std)


def trajectories_to_video(trajectories, nPosPerFrame: int, center: bool = False, image_props: Optional[dict] = None,
                          generator: Optional[torch.Generator] = None, device=None) -> torch.Tensor:
    """(N, T, 2) trajectories -> (N, T / nPosPerFrame, P, P) float32 videos (helpersGeneration.py:128-319).
    The y axis is flipped like the reference does, without mutating the caller's array."""
    props = dict(DEFAULT_IMAGE_PROPS)
    props.update(image_props or {})
    traj = _as_tensor(trajectories, device).clone()
    dev = traj.device
    traj[:, :, 1] *= -1
    if props["trajectory_unit"] != -1:
        traj = traj * props["trajectory_unit"] / (props["resolution"] * 1e9)
    N, T, _ = traj.shape
    if T % nPosPerFrame != 0:
        raise Exception("T is not divisble by posPerFrame")
    F_ = T // nPosPerFrame
    pm, ps = props["particle_intensity"]
    bm, bs = props["background_intensity"]
    if pm > 1e-4 and ps > 1e-4:
        amp = pm / nPosPerFrame + (ps / nPosPerFrame) * torch.randn(N, F_, nPosPerFrame, generator=generator, device=dev)
    else:
        amp = torch.zeros(N, F_, nPosPerFrame, device=dev)
    vid = render_frames(traj, nPosPerFrame, [psf_sigma_hr(props)], props["output_size"], props["upsampling_factor"],
                        amp, center)[:, 0]
    vid = vid + clipped_background(vid.shape, bm, bs, generator, dev)
    pn = props["poisson_noise"]
    if pn != -1:
        vid = vid * torch.poisson(torch.full(vid.shape, float(pn), device=dev), generator=generator) / pn
    return vid.float()


def normalize_images(images, background_mean=None, background_sigma=None, theoretical_max=None, clip_image=False):
    """(im - (bg_mean - bg_sigma)) / (theoretical_max - (bg_mean - bg_sigma))  (helpersGeneration.py:356-400)."""
    images = _as_tensor(images, None)
    if background_mean is None:
        background_mean = float(images.mean())
    if background_sigma is None:
        background_sigma = float(images.std(unbiased=False))
    if theoretical_max is None:
        theoretical_max = float(images.max())
    denom = theoretical_max - (background_mean - background_sigma)
    if denom == 0:
        raise ValueError("Denominator in normalization is zero. Check your inputs.")
    out = (images - (background_mean - background_sigma)) / denom
    if clip_image:
        out = out.clamp(0, 1.5)
    return out, (background_mean, background_sigma, theoretical_max)
