"""Training loop of the PSF x noise experiment: drop-in for the reference's
``Experiments/PSFNoise/trainModelsPSFNoise.py`` (data refresh per cycle :113-173, per-model epoch :177-196,
validation :206-238, ``save_results`` schema :14-22, adaptive batch doubling :117-119).

    python -m moleculardiffusion_mivit_amd.experiments.PSFNoise.trainModelsPSFNoise            # the reference run
    ... run_training(num_cycles=2, N=8, psf_indices=[0], noise_indices=[0])                   # a small one

The reference is a flat script; the same statements live in ``run_training`` so tests can drive a reduced run.
"""
import datetime

import numpy as np
import torch
from torch.utils.data import DataLoader

from ...helpers import generation as gen
from .._common import DataParallel, backward_and_step, make_scaler
from .trainSettingsPSFNoise import *       # noqa: F401,F403  (constants + factories, as the reference does :7)
from . import trainSettingsPSFNoise as S


def save_results(validation_losses, all_gen_labels, models, path_addition="", out_dir="."):
    save_path = f"{out_dir}/training_results_PSFNoise{path_addition}.pth"
    results = {"validation_losses": validation_losses, "all_labels": all_gen_labels,
               "model_weights": {name: model.state_dict() for name, model in models.items()}}
    torch.save(results, save_path)
    print(f"\nTraining results saved to {save_path}")
    return save_path


def run_training(num_cycles=100, N=64, TrainingDs_list=([1, 1], [3, 1], [5, 1], [7, 1], [9, 1], [10.2, 1]),
                 shuffle=True, verbose=False, seed=None, out_dir=".", save=True, device=None, **model_kwargs):
    device = device or S.device
    par = DataParallel()                                       # a no-op unless torch.distributed runs > 1 rank
    print = par.log                                            # noqa: A001  (rank 0 speaks)
    save = save and par.rank == 0
    print("Using device:", device)
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    models, optimizers, schedulers = S.getTrainingModels(**model_kwargs)
    for name in models:
        models[name] = models[name].to(device)
    par.prepare(models)
    batch_size = 1 if S.adaptive_batch_size != -1 else 16
    val_videos = S.load_validation_data(S.nFrames, skip_inorder=True, generator=g)[:5]
    val_labels = torch.tensor([1, 3, 5, 7, 9], dtype=torch.float32)
    validation_losses = {name: {f"val_{label.item()}": [] for label in val_labels} for name in models}
    for name in validation_losses:
        validation_losses[name]["val_avg"] = []
    all_gen_labels = np.array([])
    scalers = {name: make_scaler(model) for name, model in models.items()}
    print("StartTime: ", datetime.datetime.now())

    for cycle in range(num_cycles):
        if S.adaptive_batch_size != -1 and cycle != 0 and cycle % S.adaptive_batch_size == 0:
            batch_size *= 2
            print(f"Cycle: {cycle} new batch size: {batch_size}")
        print(f"Cycle {cycle + 1} out of {num_cycles}: {(cycle + 1) / num_cycles * 100:.2f}%")
        all_videos, all_labels = [], []
        for Ds in TrainingDs_list:
            trajs, labels = gen.brownian_single_state(N if Ds[0] != 10.2 else N // 2, S.T, Ds=Ds, alphas=1, generator=g)
            trajs, labels = trajs.permute(1, 0, 2).numpy(), labels.permute(1, 0, 2).numpy()
            all_gen_labels = np.append(all_gen_labels, labels[:, 0, 1])
            all_labels.append(labels[:, 0, 1])
            all_videos.append(S.trajs_to_vid_psf_noise(trajs / S.traj_div_factor, S.nPosPerFrame, center=S.center,
                                                       image_props=S.image_props, PSF_Settings=S.PSF_Settings,
                                                       Noise_Settings=S.Noise_Settings, generator=g))
        all_videos = torch.Tensor(np.concatenate(all_videos, axis=0))
        all_labels = torch.Tensor(np.concatenate(all_labels, axis=0) / S.D_max_normalization).unsqueeze(-1)
        dataloader = DataLoader(ImageDataset(all_videos, all_labels), batch_size=batch_size, shuffle=shuffle, generator=g)

        for name, model in models.items():
            model.train()
            optimizer, scheduler = optimizers[name], schedulers[name]
            for batch_images, batch_labels in dataloader:
                (batch_images, batch_labels), weight = par.shard(batch_images, batch_labels)
                batch_images, batch_labels = batch_images.to(device), batch_labels.to(device)
                optimizer.zero_grad()
                predictions = S.make_prediction(model, name, batch_images, eval=False)
                loss = S.loss_function(predictions, batch_labels)
                if weight != 1.0:
                    loss = loss * weight
                backward_and_step(loss, optimizer, scalers[name], model, par)
            scheduler.step()

        for name, model in models.items():
            model.eval()
            with torch.no_grad():
                label_losses = []
                for vid, label_value in zip(val_videos, val_labels):
                    label = torch.full((vid.shape[0],), label_value.item(), device=device).view(-1, 1)
                    pred = S.make_prediction(model, name, vid.to(device), True) * S.D_max_normalization
                    avg_val_loss = S.loss_function(pred, label).item()
                    validation_losses[name][f"val_{label_value.item()}"].append(avg_val_loss)
                    label_losses.append(avg_val_loss)
                    if verbose:
                        print(f"{name} on val_{label_value.item()}: Validation Loss = {avg_val_loss:.4f}")
                validation_losses[name]["val_avg"].append(float(np.mean(label_losses)))
        if save and num_cycles - cycle - 1 < 5:
            save_results(validation_losses, all_gen_labels, models, path_addition=str(num_cycles - cycle), out_dir=out_dir)

    print(f"Number of generated sequences: {all_gen_labels.shape}")
    if save:
        save_results(validation_losses, all_gen_labels, models, out_dir=out_dir)
    print(datetime.datetime.now())
    return models, validation_losses, all_gen_labels


if __name__ == "__main__":
    run_training()
