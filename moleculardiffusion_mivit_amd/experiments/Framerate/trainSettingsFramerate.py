"""Exposure-time experiment settings: drop-in for the reference's ``Experiments/Framerate/trainSettingsFramerate.py``
(constants :9-84, ``getTrainingModels`` :86-117 with keys ``tr_{i}`` / ``res_{i}``, ``make_prediction`` :157-166 slicing
``images[:, idx, :frames]``, ``trajs_to_vid_framerates`` :170-202 with zero padding to the longest sequence)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

from ...helpers.models import *            # noqa: F401,F403
from ...helpers.models import DeepResNetEmbedding, GeneralTransformer, LinearProjectionEmbedding, CNNEmbedding, MLPHead, MultiImageResNet
from ...helpers import generation as gen
from .. import _common as C

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
sequences = False
center = True
adaptive_batch_size = 20
lr = 1e-4
D_max_normalization = 10
loss_function = nn.MSELoss()
val_loss_function = nn.MSELoss(reduction='none')
single_prediction = True
use_regression_token = True
use_pos_encoding = False
tr_activation_fct = F.relu
patch_size = 13
embed_dim = 64
num_heads = 4
hidden_dim = 128
num_layers = 6
dropout = 0.0
traj_div_factor = 100
originalNposPerFrame = 10
nPosPerFrame = [5, 10, 15, 20, 30, 50]
N_POSPERFRAME = len(nPosPerFrame)
nFrames = 30
T = nFrames * originalNposPerFrame
nPosPerFrame_FramesNumber = [T // x for x in nPosPerFrame]
background_mean, background_sigma = C.BACKGROUND_MEAN, C.BACKGROUND_SIGMA
part_mean, part_std = C.PART_MEAN, C.PART_STD
image_props = C.real_data_image_props(patch_size)
_EMBEDDINGS = {"deepresnet": DeepResNetEmbedding, "linear": LinearProjectionEmbedding, "cnn": CNNEmbedding}


def getTrainingModels(lr=1e-4, embedding="deepresnet", precision=None, include_resnet=True, indices=None):
    embed_kwargs = {"patch_size": patch_size, "embed_dim": embed_dim}
    models = {}
    for i in (range(N_POSPERFRAME) if indices is None else indices):
        models[f"tr_{i}"] = GeneralTransformer(
            embedding_cls=_EMBEDDINGS[embedding], embed_kwargs=embed_kwargs, embed_dim=embed_dim, num_heads=num_heads,
            hidden_dim=hidden_dim, num_layers=num_layers, mlp_head=MLPHead, tr_activation_fct=tr_activation_fct,
            dropout=dropout, use_pos_encoding=use_pos_encoding, use_regression_token=use_regression_token,
            single_prediction=single_prediction, precision=precision)
        if include_resnet:
            models[f"res_{i}"] = MultiImageResNet(patch_size, single_prediction=single_prediction, activation=nn.ReLU)
    optimizers = {name: optim.AdamW(model.parameters(), lr=lr) for name, model in models.items()}
    schedulers = {name: optim.lr_scheduler.StepLR(opt, step_size=5, gamma=0.9) for name, opt in optimizers.items()}
    return models, optimizers, schedulers


def make_prediction(model, name, images, eval=True):
    idx = int(name.split("_")[1])
    return model(images[:, idx, :nPosPerFrame_FramesNumber[idx]])


def trajs_to_vid_framerates(trajectories, nPosPerFrame=nPosPerFrame, center=False, image_props=image_props, generator=None):
    """(N, T, 2) -> (N, len(nPosPerFrame), T // nPosPerFrame[0], P, P): one rendering per exposure setting (particle flux
    scaled with the exposure), normalised, zero-padded to the longest sequence."""
    trajectories = torch.as_tensor(np.asarray(trajectories), dtype=torch.float32)
    N, T_, _ = trajectories.shape
    max_frames = T_ // nPosPerFrame[0]
    part_flux, pstd = image_props["particle_intensity"]
    bg_mean, bg_sigma = image_props["background_intensity"][:2]
    out = torch.zeros((N, len(nPosPerFrame), max_frames, patch_size, patch_size))
    for i, sub in enumerate(nPosPerFrame):
        if T_ % sub != 0:
            raise Exception("T is not divisible by nPosPerFrame")
        flux = part_flux * (sub / originalNposPerFrame)
        props = dict(image_props)
        props["particle_intensity"] = [flux, pstd]
        vid = gen.trajectories_to_video(trajectories, sub, center=center, image_props=props, generator=generator)
        vid, _ = gen.normalize_images(vid, bg_mean, bg_sigma, bg_mean + flux)
        out[:, i, :T_ // sub] = vid
    return out


val_d_in_order = np.arange(0.1, 10.01, 0.1)
N_in_order = 10


def load_validation_data(length=20, skip_inorder=False, generator=None, n_synthetic=50):
    g = generator or torch.Generator().manual_seed(20250815)
    sets, tio = C.validation_trajectories(length, T, traj_div_factor, g, n_synthetic,
                                          None if skip_inorder else (val_d_in_order, N_in_order))
    vids = [trajs_to_vid_framerates(t, nPosPerFrame, center=center, image_props=image_props, generator=g) for t in sets]
    vio = torch.zeros(1) if skip_inorder else trajs_to_vid_framerates(tio, nPosPerFrame, center=center,
                                                                     image_props=image_props, generator=g)
    return (*vids, vio)
