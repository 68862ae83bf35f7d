"""PSF x noise experiment settings: drop-in for the reference's ``Experiments/PSFNoise/trainSettingsPSFNoise.py``
(same constant names :9-85, ``getTrainingModels`` :90-125, ``load_validation_data`` :131-160, ``make_prediction``
:164-172, ``select_models_from_psf/noise`` :175-193, ``trajs_to_vid_psf_noise`` :196-309) with the HIP-backed model
classes and a seedable, dependency-free video synthesiser.

Additions the reference does not have (all optional, defaults reproduce the reference):
``getTrainingModels(..., embedding=, precision=, include_resnet=, psf_indices=, noise_indices=)`` and an RNG
``generator`` argument on the data functions; ``MIVIT_VALIDATION_ROOT`` points at the reference's
``Experiments/validation_trajectories`` folder (synthetic Brownian validation sets are used when it is absent).
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

from ...helpers.models import *            # noqa: F401,F403  (same star-import surface as the reference :4)
from ...helpers.models import (DeepResNetEmbedding, GeneralTransformer, LinearProjectionEmbedding, CNNEmbedding,
                               MLPHead, MultiImageResNet)
from ...helpers import generation as gen

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")

sequences = False
center = True
adaptive_batch_size = 20          # batch size doubles every `adaptive_batch_size` cycles (-1: fixed)
lr = 1e-4
D_max_normalization = 10

loss_function = nn.MSELoss()
val_loss_function = nn.MSELoss(reduction='none')
single_prediction = True
use_regression_token = True
use_pos_encoding = False
tr_activation_fct = F.relu

patch_size = 9
embed_dim = 64
num_heads = 4
hidden_dim = 128
num_layers = 6
dropout = 0.0

traj_div_factor = 100             # trajectories are given in pixels/s, wanted in the ms domain
nPosPerFrame = 10
nFrames = 30
T = nFrames * nPosPerFrame

PSF_Settings = [2, 1.75, 1.5, 1.25, 1]
Noise_Settings = [0, 1 / 50, 1 / 25, 1 / 20, 1 / 10, 1 / 5]
N_PSF, N_Noise = len(PSF_Settings), len(Noise_Settings)

background_mean = 5000
part_mean, part_std = 5000, 500

image_props = {
    "particle_intensity": [part_mean, part_std],
    "NA": 1.46,
    "wavelength": 500e-9,
    "psf_division_factor": 1.3,
    "resolution": 100e-9,
    "output_size": patch_size,
    "upsampling_factor": 5,
    "background_intensity": [background_mean, 0],
    "poisson_noise": 100,
    "trajectory_unit": 1200,
}

_EMBEDDINGS = {"deepresnet": DeepResNetEmbedding, "linear": LinearProjectionEmbedding, "cnn": CNNEmbedding}


def getTrainingModels(lr=1e-4, embedding="deepresnet", precision=None, include_resnet=True, psf_indices=None,
                      noise_indices=None):
    """{'tr_{psf}_{noise}': MiViT, 'res_{psf}_{noise}': ResNet baseline}, one AdamW + StepLR(5, 0.9) per model."""
    embed_kwargs = {"patch_size": patch_size, "embed_dim": embed_dim}
    models = {}
    for psf_index in (range(N_PSF) if psf_indices is None else psf_indices):
        for noise_index in (range(N_Noise) if noise_indices is None else noise_indices):
            models[f"tr_{psf_index}_{noise_index}"] = GeneralTransformer(
                embedding_cls=_EMBEDDINGS[embedding], embed_kwargs=embed_kwargs, embed_dim=embed_dim,
                num_heads=num_heads, hidden_dim=hidden_dim, num_layers=num_layers, mlp_head=MLPHead,
                tr_activation_fct=tr_activation_fct, dropout=dropout, use_pos_encoding=use_pos_encoding,
                use_regression_token=use_regression_token, single_prediction=single_prediction, precision=precision)
            if include_resnet:
                models[f"res_{psf_index}_{noise_index}"] = MultiImageResNet(
                    patch_size, single_prediction=single_prediction, activation=nn.ReLU)
    optimizers = {name: optim.AdamW(model.parameters(), lr=lr) for name, model in models.items()}
    schedulers = {name: optim.lr_scheduler.StepLR(opt, step_size=5, gamma=0.9) for name, opt in optimizers.items()}
    return models, optimizers, schedulers


val_d_in_order = np.arange(0.1, 10.01, 0.1)
N_in_order = 10


def _validation_root():
    for cand in (os.environ.get("MIVIT_VALIDATION_ROOT"), "../validation_trajectories"):
        if cand and os.path.isdir(cand):
            return cand
    return None


def load_validation_data(length=20, skip_inorder=False, generator=None, n_synthetic=50):
    """Five fixed validation sets (D = 1, 3, 5, 7, 9) rendered for every PSF x noise cell, + the in-order set.
    Reads the reference's ``validation_trajectories/<length>/val{D}.npy`` when available, otherwise draws seeded
    Brownian trajectories with those D."""
    if length not in (20, 30):
        raise ValueError("Invalid length value, select one in: [20, 30]")
    root = _validation_root()
    g = generator or torch.Generator().manual_seed(20250815)
    vids = []
    for D in (1, 3, 5, 7, 9):
        if root is not None:
            trajs = np.load(os.path.join(root, str(length), f"val{D}.npy")) / traj_div_factor
        else:
            tr, _ = gen.brownian_single_state(n_synthetic, T, Ds=[D, 0.0], generator=g)
            trajs = tr.permute(1, 0, 2).numpy() / traj_div_factor
        vids.append(torch.as_tensor(trajs_to_vid_psf_noise(trajs, nPosPerFrame, center=center, image_props=image_props,
                                                          PSF_Settings=PSF_Settings, Noise_Settings=Noise_Settings,
                                                          generator=g)))
    if skip_inorder:
        vid_inorder = torch.zeros(1)
    else:
        if root is not None:
            tio = (np.load(os.path.join(root, "valTrajsInOrder.npy")) / traj_div_factor).reshape(-1, T, 2)
        else:
            parts = [gen.brownian_single_state(N_in_order, T, Ds=[float(D), 0.0], generator=g)[0].permute(1, 0, 2)
                     for D in val_d_in_order]
            tio = torch.cat(parts).numpy() / traj_div_factor
        vio = trajs_to_vid_psf_noise(tio, nPosPerFrame, center=center, image_props=image_props,
                                     PSF_Settings=PSF_Settings, Noise_Settings=Noise_Settings, generator=g)
        vid_inorder = torch.as_tensor(vio).reshape(len(val_d_in_order), N_in_order, N_PSF, N_Noise, nFrames, patch_size,
                                                   patch_size)
    return (*vids, vid_inorder)


def make_prediction(model, name, images, eval=True):
    prefix, psf_index, noise_index = name.split("_")
    return model(images[:, int(psf_index), int(noise_index)])


def select_models_from_psf(models, wanted_psf_index, wanted_prefix=None):
    return [n for n in models if int(n.split("_")[1]) == wanted_psf_index
            and (wanted_prefix is None or n.split("_")[0] == wanted_prefix)]


def select_models_from_noise(models, wanted_noise_index, wanted_prefix=None):
    return [n for n in models if int(n.split("_")[2]) == wanted_noise_index
            and (wanted_prefix is None or n.split("_")[0] == wanted_prefix)]


def trajs_to_vid_psf_noise(trajectories, nPosPerFrame, center=False, image_props={}, PSF_Settings=[], Noise_Settings=[],
                           generator=None, device="cpu", reference_chain=True):
    """(N, T, 2) trajectories -> (N, N_PSF, N_Noise, nFrames, P, P) float32: one noise-free rendering per PSF width
    (sigma / PSF_Settings[i]), one frame intensity ~ N(part_mean, part_std) shared by its sub-positions, then per
    noise level a clipped-Gaussian background (std = part_mean * level) and Poisson(frame * pn) / pn.

    ``reference_chain=True`` (default) reproduces the reference's loop exactly (trainSettingsPSFNoise.py:296-306): the
    level-0 result Poisson(clean + background) OVERWRITES out[psf, 0, f], and every level j >= 1 is then built on that
    already-noised frame -- background added twice (~2 * bm), Poisson applied twice.  That is the input distribution the
    reference's checkpoints and published validation losses were produced on.  ``reference_chain=False`` builds every
    level from the clean frame (one background, one Poisson draw): the cleaner camera model, NOT comparable with the
    reference's numbers."""
    if len(PSF_Settings) == 0 or len(Noise_Settings) == 0:
        raise Exception("No settings given")
    props = dict(gen.DEFAULT_IMAGE_PROPS)
    props["poisson_noise"] = 1                      # default of the reference's PSFNoise renderer (:231)
    props.update(image_props)
    traj = torch.as_tensor(np.asarray(trajectories), dtype=torch.float32, device=device)
    N, T_, _ = traj.shape
    if T_ % nPosPerFrame != 0:
        raise Exception("T is not divisble by posPerFrame")
    if props["trajectory_unit"] != -1:
        traj = traj * props["trajectory_unit"] * 1e-9 / props["resolution"]
    F_ = T_ // nPosPerFrame
    pm, ps = props["particle_intensity"]
    bm = props["background_intensity"][0]
    # NB the reference's PSFNoise renderer ignores psf_division_factor (:249): fwhm = wavelength / 2 * NA
    sigma = props["upsampling_factor"] / props["resolution"] * (props["wavelength"] / 2 * props["NA"]) / 2.355
    frame_int = pm + ps * torch.randn(N, F_, 1, generator=generator, device=traj.device)
    amp = (frame_int / nPosPerFrame).expand(N, F_, nPosPerFrame) if (pm > 1e-4 and ps > 1e-4) else torch.zeros(N, F_, nPosPerFrame)
    clean = gen.render_frames(traj, nPosPerFrame, [sigma / s for s in PSF_Settings], props["output_size"],
                              props["upsampling_factor"], amp, center)           # (N, N_PSF, F, P, P)
    pn = props["poisson_noise"]
    out = torch.empty(N, len(PSF_Settings), len(Noise_Settings), F_, props["output_size"], props["output_size"])
    base = clean
    for j, level in enumerate(Noise_Settings):
        noisy = base + gen.clipped_background(clean.shape, bm, part_mean * level, generator, traj.device)
        frame = torch.poisson((noisy * pn).clamp_min(0), generator=generator) / pn
        out[:, :, j] = frame.cpu()
        if reference_chain and j == 0:
            base = frame                              # the reference reads out_video[psf, 0, f] back for the later levels
    return out.numpy()
