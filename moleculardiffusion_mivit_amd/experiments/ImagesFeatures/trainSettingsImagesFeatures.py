"""Images + hand-crafted features experiment settings: drop-in for the model side of the reference's
``Experiments/ImagesFeatures/trainSettingsImagesFeatures.py`` (constants :9-92, ``getTrainingModels`` :112-188 with
keys im_tr / im_ft_late_tr / im_ft_early_tr / im_resnet / im_ft_resnet / ft_mlp, rotation test-time augmentation
:255-300, name-dispatched ``make_prediction`` :303-341).

The 25 trajectory descriptors (``helpers/features.compute_diffusion_features``, pinned against the reference's
``helpers/helpersFeatures.py``) are a CPU pre-processing step outside the hot path: the model sees a
``[B, 25]`` tensor.  ``create_video_and_feature_pairs`` / ``load_validation_data`` mirror helpersGeneration.py:674-720 and
trainSettingsImagesFeatures.py:194-231."""
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

import numpy as np

from ...helpers import features as ft
from ...helpers import generation as gen
from ...helpers.models import *            # noqa: F401,F403
from ...helpers.models import DeepResNetEmbedding, GeneralTransformer, MLPHead, MultiImageFeatureResNet, MultiImageResNet
from .. import _common as C

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
sequences = False
center = True
adaptive_batch_size = 20
lr = 1e-4
D_max_normalization = 10
msdPerfect, msdFrame, msdLocalized = ("MSD_Perfect", "MSD_Frame", "MSD_Localized")
MSDModels = [msdPerfect, msdFrame, msdLocalized]
MSD_mult_factor = 250
MSD_mult_factor_avg = 37.5
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
dt = 1
traj_div_factor = 100
nPosPerFrame = 10
nFrames = 30
T = nFrames * nPosPerFrame
N_features = 25                            # helpers/helpersFeatures.py:34
background_mean, background_sigma = C.BACKGROUND_MEAN, C.BACKGROUND_SIGMA
part_mean, part_std = C.PART_MEAN, C.PART_STD
image_props = C.real_data_image_props(patch_size)

localization_uncertainty = (0, 0)
val_d_in_order = np.arange(0.1, 10.01, 0.1)
N_in_order = 10

im_resnet, im_ft_resnet = "im_resnet", "im_ft_resnet"
ft_mlp = "ft_mlp"
im_tr, im_ft_early_tr, im_ft_late_tr = "im_tr", "im_ft_early_tr", "im_ft_late_tr"


def getTrainingModels(lr=1e-4, addMSDModels=False, embedding_cls=DeepResNetEmbedding, precision=None):
    embed_kwargs = {"patch_size": patch_size, "embed_dim": embed_dim}
    common = dict(embedding_cls=embedding_cls, embed_kwargs=embed_kwargs, embed_dim=embed_dim, num_heads=num_heads,
                  hidden_dim=hidden_dim, num_layers=num_layers, mlp_head=MLPHead, tr_activation_fct=tr_activation_fct,
                  dropout=dropout, use_pos_encoding=use_pos_encoding, use_regression_token=use_regression_token,
                  single_prediction=single_prediction, precision=precision)
    models = {
        im_tr: GeneralTransformer(**common),
        im_ft_late_tr: GeneralTransformer(**common, use_global_features=True, fusion_type='late', global_feature_dim=N_features),
        im_ft_early_tr: GeneralTransformer(**common, use_global_features=True, fusion_type='early', global_feature_dim=N_features),
        im_resnet: MultiImageResNet(patch_size, single_prediction=single_prediction, activation=nn.ReLU),
        im_ft_resnet: MultiImageFeatureResNet(patch_size, N_features, feature_size=embed_dim, hidden_size=hidden_dim,
                                              activation=nn.ReLU),
        ft_mlp: MLPHead(input_dim=N_features),
    }
    optimizers = {name: optim.AdamW(model.parameters(), lr=lr) for name, model in models.items()}
    schedulers = {name: optim.lr_scheduler.StepLR(opt, step_size=5, gamma=0.9) for name, opt in optimizers.items()}
    if addMSDModels:
        models.update({m: None for m in MSDModels})
    return models, optimizers, schedulers


def generate_rotated_sequences(x):
    if x.ndim != 4:
        raise ValueError(f"Expected tensor of shape (B, T, H, W), got {x.shape}")
    return tuple(torch.rot90(x, k=k, dims=(2, 3)) if k else x for k in range(4))


def predict_with_rotations(model, images, features=None):
    """Mean prediction over the 0 / 90 / 180 / 270 degree rotations of every frame (test-time augmentation)."""
    preds = [model(r.contiguous(), features) if features is not None else model(r.contiguous())
             for r in generate_rotated_sequences(images)]
    return torch.stack(preds, dim=0).mean(dim=0)


def d_fromMSDTau1(trajectories):
    """Mean squared displacement at lag 1 per trajectory ((N, T, 2) -> (N,))."""
    d = trajectories[:, 1:] - trajectories[:, :-1]
    return (d ** 2).sum(-1).mean(dim=1)


def make_prediction(model, name, images, features, trajectories=None, msd_mult_fact=MSD_mult_factor, eval=True):
    images, features = images.to(device), features.to(device)
    if "MSD" not in name and eval:
        model.eval()
    rot = "rot" in name
    if name.startswith("im_resnet"):
        return predict_with_rotations(model, images) if rot else model(images)
    if name.startswith("im_ft_resnet"):
        return predict_with_rotations(model, images, features) if rot else model(images, features)
    if name == ft_mlp:
        return model(features)
    if name == msdPerfect:
        return d_fromMSDTau1(trajectories[0]) * msd_mult_fact
    if name == msdFrame:
        return d_fromMSDTau1(trajectories[1]) * MSD_mult_factor_avg
    if name == msdLocalized:
        return d_fromMSDTau1(trajectories[2]) * MSD_mult_factor_avg
    with torch.no_grad():
        feats = features if "ft" in name else None
        return predict_with_rotations(model, images, feats) if rot else model(images, feats)


def create_video_and_feature_pairs(trajectories, nPosPerFrame, center, image_props, localization_uncertainty=(0, 0), dt=1.0,
                                   generator=None):
    """(N, T, 2) trajectories -> (normalised videos [N, nFrames, P, P], features [N, 25], (trajectories, frame-averaged,
    frame-averaged + localisation error)) -- helpersGeneration.py:674-720."""
    traj = np.asarray(trajectories, dtype=np.float32)
    bg_mean, bg_sigma = image_props["background_intensity"]
    pm = image_props["particle_intensity"][0]
    vid = gen.trajectories_to_video(torch.as_tensor(traj), nPosPerFrame, center=center, image_props=image_props, generator=generator)
    vid = gen.normalize_images(vid, bg_mean, bg_sigma, pm + bg_mean)[0]
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,), generator=generator)) if generator is not None else None
    feats, avg, noisy = ft.compute_features_for_trajectories(traj, nPosPerFrame, dt=dt, localization_uncertainty=localization_uncertainty,
                                                             rng=np.random.default_rng(seed))
    return np.asarray(vid, dtype=np.float32), feats.astype(np.float32), (traj, avg, noisy)


def load_validation_data(length=20, skip_inorder=False, generator=None, n_synthetic=50):
    """((videos, features, trajectory triple) for D = 1, 3, 5, 7, 9, and the in-order set) -- reference :194-231.  Uses the
    reference's validation_trajectories/*.npy when present (MIVIT_VALIDATION_ROOT), seeded Brownian sets otherwise."""
    g = generator or torch.Generator().manual_seed(20250815)
    sets, tio = C.validation_trajectories(length, T, traj_div_factor, g, n_synthetic,
                                          None if skip_inorder else (val_d_in_order, N_in_order))
    out = []
    for tr in sets:
        v, f, t = create_video_and_feature_pairs(tr, nPosPerFrame, center, image_props, localization_uncertainty, dt, g)
        out.append((torch.Tensor(v), torch.Tensor(f), t))
    if skip_inorder:
        out.append((torch.zeros(1), torch.zeros(1), np.zeros(1)))
    else:
        v, f, t = create_video_and_feature_pairs(tio, nPosPerFrame, center, image_props, localization_uncertainty, dt, g)
        out.append((torch.Tensor(v).reshape(len(val_d_in_order), N_in_order, nFrames, patch_size, patch_size),
                    torch.Tensor(f).reshape(len(val_d_in_order), N_in_order, N_features), t))
    return tuple(out)


def make_prediction_tuple(model, name, vid_ft_trajs, eval=True):
    """make_prediction on a (videos, features, trajectories) triple (reference trainSettingsImagesFeatures.py:343-345)."""
    return make_prediction(model, name, vid_ft_trajs[0], vid_ft_trajs[1], vid_ft_trajs[2], eval=eval)
