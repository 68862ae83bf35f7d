"""Training loop of the images + features experiment: drop-in for ``Experiments/ImagesFeatures/trainModelsImagesFeatures.py``
(:112-255): per cycle fresh Brownian trajectories at D = 1, 3, 5, 7, 9 -> (normalised videos, 25 features) pairs ->
ImageFeatureDataset -> every model of the zoo trained on the same minibatches, dispatched by name exactly as the
reference's inner loop (:184-195: im_resnet sees images, ft_mlp features, names without "ft" no features) -> StepLR ->
validation on D = 1..9 with predictions * D_max_normalization -> save_results("training_results_ten*.pth", :20-24)."""
import numpy as np
import torch

from ...helpers import generation as gen
from .. import _common as C
from . import trainSettingsImagesFeatures as S
from .trainSettingsImagesFeatures import *     # noqa: F401,F403


def predict_train(model, name, images, features):
    """The reference's training-time dispatch (trainModelsImagesFeatures.py:184-195)."""
    if name == S.im_resnet:
        return model(images)
    if name == S.ft_mlp:
        return model(features)
    return model(images, features if "ft" in name else None)


def run_training(num_cycles=100, N=64, TrainingDs_list=([1, 1], [3, 1], [5, 1], [7, 1], [9, 1]), seed=None, out_dir=".",
                 save=True, device=None, verbose=False, model_filter=None, **model_kwargs):
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    models, optimizers, schedulers = S.getTrainingModels(**model_kwargs)
    if model_filter is not None:
        models = {k: v for k, v in models.items() if k in model_filter}
    vals = S.load_validation_data(S.nFrames, skip_inorder=True, generator=g)[:5]
    val_sets = [((v, f), D) for (v, f, _), D in zip(vals, C.D_VALUES)]

    def make_batch_data(cycle):
        vids, feats, labs = [], [], []
        for Ds in TrainingDs_list:
            trajs, labels = gen.brownian_single_state(N, S.T, Ds=Ds, alphas=1, generator=g)
            labs.append(labels[0, :, 1].numpy())
            v, f, _ = S.create_video_and_feature_pairs(trajs.permute(1, 0, 2).numpy() / S.traj_div_factor, S.nPosPerFrame, S.center,
                                                       S.image_props, generator=g)
            vids.append(torch.as_tensor(v))
            feats.append(torch.as_tensor(f))
        raw = np.concatenate(labs)
        return (torch.cat(vids).float(), torch.cat(feats).float(),
                torch.tensor(raw / S.D_max_normalization, dtype=torch.float32).unsqueeze(-1), raw)

    def predict(model, name, images, features):
        return predict_train(model, name, images, features) if model.training else S.make_prediction(model, name, images, features)

    return C.run_cycles(S, models, optimizers, schedulers, make_batch_data, predict, num_cycles, val_sets, "ten",
                        device=device, out_dir=out_dir, save=save, generator=g, verbose=verbose)


if __name__ == "__main__":
    run_training()
