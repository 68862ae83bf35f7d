"""Training loop of the embedding ablation: drop-in for ``Experiments/Embeddings/trainModelsEmbeddings.py`` (:117-244;
the models are called directly, ``model(batch_images)`` :196)."""
import numpy as np
import torch

from ...helpers import generation as gen
from .. import _common as C
from . import trainSettingsEmbeddings as S
from .trainSettingsEmbeddings import *     # noqa: F401,F403


def run_training(num_cycles=100, N=64, TrainingDs_list=([1, 1], [3, 1], [5, 1], [7, 1], [9, 1], [10.2, 1]), seed=None,
                 out_dir=".", save=True, device=None, verbose=False, model_filter=None, **model_kwargs):
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    models, optimizers, schedulers = S.getTrainingModels(**model_kwargs)
    if model_filter is not None:
        models = {k: v for k, v in models.items() if k in model_filter}
    vals = S.load_validation_data(S.nFrames, generator=g)[:5]
    val_sets = [((v,), D) for v, D in zip(vals, C.D_VALUES)]

    def make_batch_data(cycle):
        vids, labs = [], []
        for Ds in TrainingDs_list:
            trajs, labels = gen.brownian_single_state(N if Ds[0] != 10.2 else N // 2, S.T, Ds=Ds, alphas=1, generator=g)
            labs.append(labels[0, :, 1].numpy())
            vids.append(S.render(trajs.permute(1, 0, 2) / S.traj_div_factor, g))
        raw = np.concatenate(labs)
        return torch.cat(vids).float(), torch.tensor(raw / S.D_max_normalization, dtype=torch.float32).unsqueeze(-1), raw

    predict = lambda model, name, images: model(images)   # noqa: E731
    return C.run_cycles(S, models, optimizers, schedulers, make_batch_data, predict, num_cycles, val_sets, "embed",
                        device=device, out_dir=out_dir, save=save, generator=g, verbose=verbose)


if __name__ == "__main__":
    run_training()
