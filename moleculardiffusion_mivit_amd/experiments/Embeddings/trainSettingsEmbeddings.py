"""Embedding-ablation experiment settings: drop-in for the reference's
``Experiments/Embeddings/trainSettingsEmbeddings.py`` (constants :9-78, ``getTrainingModels`` :84-101 -- linear / cnn /
deepcnn embeddings x normal / small (_s) / big (_b) transformer + one ResNet --, ``load_validation_data`` :106-150,
``get_transformer_models`` :152-211).  ``use_pos_encoding`` is True here, as in the reference."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

from ...helpers.models import *            # noqa: F401,F403
from ...helpers.models import (CNNEmbedding, DeepResNetEmbedding, GeneralTransformer, LinearProjectionEmbedding, MLPHead,
                               MultiImageResNet)
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
use_pos_encoding = True
tr_activation_fct = F.relu
patch_size = 9
embed_dim = 64
num_heads = 4
hidden_dim = 128
num_layers = 6
dropout = 0.0
traj_div_factor = 100
nPosPerFrame = 10
nFrames = 30
T = nFrames * nPosPerFrame
background_mean, background_sigma = C.BACKGROUND_MEAN, C.BACKGROUND_SIGMA
part_mean, part_std = C.PART_MEAN, C.PART_STD
image_props = C.real_data_image_props(patch_size)


def get_transformer_models(patch_size=patch_size, embed_dim=embed_dim, num_heads=num_heads, hidden_dim=hidden_dim,
                           num_layers=num_layers, dropout=dropout, use_pos_encoding=False, tr_activation_fct=F.gelu,
                           use_regression_token=True, single_prediction=True, name_suffix='', precision=None):
    """{'linear'+s, 'cnn'+s, 'deepcnn'+s}: the same transformer behind the three frame embeddings."""
    embed_kwargs = {"patch_size": patch_size, "embed_dim": embed_dim}
    out = {}
    for key, emb in (("linear", LinearProjectionEmbedding), ("cnn", CNNEmbedding), ("deepcnn", DeepResNetEmbedding)):
        out[key + name_suffix] = GeneralTransformer(
            embedding_cls=emb, embed_kwargs=embed_kwargs, embed_dim=embed_dim, num_heads=num_heads, hidden_dim=hidden_dim,
            num_layers=num_layers, mlp_head=MLPHead, tr_activation_fct=tr_activation_fct, dropout=dropout,
            use_pos_encoding=use_pos_encoding, use_regression_token=use_regression_token,
            single_prediction=single_prediction, precision=precision)
    return out


def getTrainingModels(lr=1e-4, precision=None):
    kw = dict(dropout=dropout, use_pos_encoding=use_pos_encoding, tr_activation_fct=tr_activation_fct,
              use_regression_token=use_regression_token, single_prediction=single_prediction, precision=precision)
    models = get_transformer_models(patch_size, embed_dim, num_heads, hidden_dim, num_layers, name_suffix='_n', **kw)
    models.update(get_transformer_models(patch_size, embed_dim // 2, num_heads // 2, hidden_dim // 2, num_layers // 2,
                                         name_suffix='_s', **kw))
    models.update(get_transformer_models(patch_size, embed_dim * 2, num_heads * 2, hidden_dim * 2, num_layers * 2,
                                         name_suffix='_b', **kw))
    models["resnet"] = MultiImageResNet(patch_size, single_prediction=single_prediction, activation=nn.ReLU)
    optimizers = {name: optim.AdamW(model.parameters(), lr=lr) for name, model in models.items()}
    schedulers = {name: optim.lr_scheduler.StepLR(opt, step_size=5, gamma=0.9) for name, opt in optimizers.items()}
    return models, optimizers, schedulers


val_d_in_order = np.arange(0.1, 10.01, 0.1)
N_in_order = 10


def render(trajs, generator=None):
    """trajectories -> normalised videos, as the reference does for training and validation data (:121-123)."""
    vid = gen.trajectories_to_video(trajs, nPosPerFrame, center=True, image_props=image_props, generator=generator)
    return gen.normalize_images(vid, background_mean, background_sigma, part_mean + background_mean)[0]


def load_validation_data(length=20, generator=None, n_synthetic=50):
    g = generator or torch.Generator().manual_seed(20250815)
    sets, tio = C.validation_trajectories(length, T, traj_div_factor, g, n_synthetic, (val_d_in_order, N_in_order))
    vids = [render(t, g) for t in sets]
    vio = render(tio, g).reshape(len(val_d_in_order), N_in_order, nFrames, patch_size, patch_size)
    return (*vids, vio)
