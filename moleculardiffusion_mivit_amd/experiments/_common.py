"""Pieces shared by the experiment mirrors: the constants every reference ``trainSettings*.py`` repeats, validation
sets (reference .npy files when ``MIVIT_VALIDATION_ROOT`` / ``../validation_trajectories`` exists, seeded Brownian
otherwise), and the cycle loop every reference ``trainModels*.py`` repeats (data refresh -> per-model epoch ->
StepLR -> validation at D = 1,3,5,7,9 with predictions * D_max_normalization -> save_results)."""
import datetime
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.data import DataLoader

from ..helpers import generation as gen

D_VALUES = (1, 3, 5, 7, 9)

# values from real data, shared by Framerate / Embeddings / ImagesFeatures (e.g. trainSettingsEmbeddings.py:52-53)
BACKGROUND_MEAN, BACKGROUND_SIGMA = 1420, 290
PART_MEAN, PART_STD = 6000 - BACKGROUND_MEAN, 500


def real_data_image_props(patch_size):
    return {"particle_intensity": [PART_MEAN, PART_STD], "NA": 1.46, "wavelength": 500e-9, "psf_division_factor": 1.3,
            "resolution": 100e-9, "output_size": patch_size, "upsampling_factor": 5,
            "background_intensity": [BACKGROUND_MEAN, BACKGROUND_SIGMA], "poisson_noise": 100, "trajectory_unit": 1200}


def validation_root():
    for cand in (os.environ.get("MIVIT_VALIDATION_ROOT"), "../validation_trajectories"):
        if cand and os.path.isdir(cand):
            return cand
    return None


def validation_trajectories(length, T, traj_div_factor, generator, n_synthetic=50, in_order=None):
    """[(N,T,2) array for D in 1,3,5,7,9] (+ the in-order set when `in_order` = (d_values, n_per_d))."""
    root = validation_root()
    sets = []
    for D in D_VALUES:
        if root is not None:
            sets.append(np.load(os.path.join(root, str(length), f"val{D}.npy")) / traj_div_factor)
        else:
            tr, _ = gen.brownian_single_state(n_synthetic, T, Ds=[D, 0.0], generator=generator)
            sets.append(tr.permute(1, 0, 2).numpy() / traj_div_factor)
    tio = None
    if in_order is not None:
        dvals, n = in_order
        if root is not None:
            tio = (np.load(os.path.join(root, "valTrajsInOrder.npy")) / traj_div_factor).reshape(-1, T, 2)
        else:
            tio = torch.cat([gen.brownian_single_state(n, T, Ds=[float(d), 0.0], generator=generator)[0].permute(1, 0, 2)
                             for d in dvals]).numpy() / traj_div_factor
    return sets, tio


def make_scaler(model):
    """fp16 mode (BASELINE config 5) trains under dynamic loss scaling: init 2**16, x2 every 2000 good steps, /2 on
    inf/NaN (skipping that step).  fp32 / bf16 models need none (returns None)."""
    if getattr(model, "precision", None) != "fp16":
        return None
    return torch.amp.GradScaler("cuda", init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000)


class DataParallel:
    """Data parallelism for the training loops, active when torch.distributed is initialised with
    more than one rank (one process per GPU, `torchrun`-style); a no-op otherwise, so the single-process loops are the
    reference's.  Every rank runs the same loop on the same generated data (same seed) and takes its contiguous slice of
    each minibatch, so the job's minibatch IS the reference's; gradients are averaged over ranks (MiViT models: overlapped
    with the native backward, `dp.attach`; a DeepResNet embedding normalises with job-wide BatchNorm statistics), the
    loss of a ragged tail is weighted by its share of the minibatch, and only rank 0 prints and saves."""

    def __init__(self):
        import torch.distributed as dist
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1

    def prepare(self, models):
        # the experiment loops step at the reference's batch sizes (1 ... 256 sequences: launch-bound, host time decides) and follow
        # the pattern direct parameter gradients are made for -- optimizer.zero_grad() (grads set to None), backward, step --, so
        # every native model gets them (bitwise the same training trajectory; any other pattern falls back by itself)
        for model in models.values():
            if model is not None and hasattr(model, "direct_param_grads"):
                model.direct_param_grads(True)
        if not self.on:
            return
        from .. import dp
        for model in models.values():
            if model is None:
                continue
            if hasattr(model, "_plan"):
                dp.attach(model, sync_batchnorm=hasattr(getattr(model, "embedding", None), "sync_batchnorm"))
            else:
                dp.broadcast_parameters(model)            # stock-PyTorch baselines: averaged after backward

    def shard(self, *tensors):
        """This rank's slice of a minibatch and the weight of its mean loss in the minibatch's mean loss (x world: the
        gradient average divides by world).  A minibatch smaller than the world is computed whole on every rank."""
        n = tensors[0].shape[0]
        if not self.on or n < self.world:
            return tensors, 1.0
        lo, hi = self.rank * n // self.world, (self.rank + 1) * n // self.world
        return tuple(t[lo:hi] for t in tensors), (hi - lo) * self.world / n

    def finish(self, model):
        if self.on:
            from .. import dp
            dp.finish_external_grads(model)

    def log(self, *args, **kwargs):
        if self.rank == 0:
            print(*args, **kwargs)


def backward_and_step(loss, optimizer, scaler=None, model=None, parallel=None):
    """loss.backward(); optimizer.step()  (trainModelsPSFNoise.py:192-193), through the loss scaler when there is one;
    under data parallelism the gradients that are not reduced during backward are averaged before the step."""
    if scaler is None:
        loss.backward()
        if parallel is not None and model is not None:
            parallel.finish(model)
        optimizer.step()
    else:
        scaler.scale(loss).backward()
        if parallel is not None and model is not None:
            parallel.finish(model)
        scaler.step(optimizer)
        scaler.update()


def run_cycles(S, models, optimizers, schedulers, make_batch_data, predict, num_cycles, val_sets, results_name,
               batch_size=None, device=None, out_dir=".", save=True, shuffle=True, generator=None, verbose=False):
    """The reference's cycle loop.  `make_batch_data(cycle)` -> (tensors..., labels, raw_labels);
    `predict(model, name, *batch_tensors)` -> predictions; `val_sets` = list of (tensors tuple, D value)."""
    device = device or S.device
    for name in models:
        if models[name] is not None:
            models[name] = models[name].to(device)
    par = DataParallel()
    par.prepare(models)
    print = par.log                                            # noqa: A001  (rank 0 speaks)
    save = save and par.rank == 0
    if batch_size is None:
        batch_size = 1 if S.adaptive_batch_size != -1 else 16
    validation_losses = {name: {**{f"val_{float(D)}": [] for D in D_VALUES}, "val_avg": []} for name in models}
    all_gen_labels = np.array([])
    scalers = {}
    print("StartTime: ", datetime.datetime.now())
    for cycle in range(num_cycles):
        if S.adaptive_batch_size != -1 and cycle != 0 and cycle % S.adaptive_batch_size == 0:
            batch_size *= 2
            print(f"Cycle: {cycle} new batch size: {batch_size}")
        print(f"Cycle {cycle + 1} out of {num_cycles}: {(cycle + 1) / num_cycles * 100:.2f}%")
        *tensors, labels, raw = make_batch_data(cycle)
        all_gen_labels = np.append(all_gen_labels, raw)
        ds = torch.utils.data.TensorDataset(*tensors, labels)
        loader = DataLoader(ds, batch_size=batch_size, shuffle=shuffle, generator=generator)
        for name, model in models.items():
            if model is None:
                continue
            model.train()
            opt, sch = optimizers[name], schedulers[name]
            if name not in scalers:
                scalers[name] = make_scaler(model)
            for *bt, bl in loader:
                (*bt, bl), weight = par.shard(*bt, bl)
                bt = [t.to(device) for t in bt]
                opt.zero_grad()
                loss = S.loss_function(predict(model, name, *bt), bl.to(device))
                if weight != 1.0:
                    loss = loss * weight
                backward_and_step(loss, opt, scalers[name], model, par)
            sch.step()
        for name, model in models.items():
            if model is None:
                continue
            model.eval()
            with torch.no_grad():
                per = []
                for vt, D in val_sets:
                    vt = [t.to(device) for t in vt]
                    label = torch.full((vt[0].shape[0], 1), float(D), device=device)
                    v = S.loss_function(predict(model, name, *vt) * S.D_max_normalization, label).item()
                    validation_losses[name][f"val_{float(D)}"].append(v)
                    per.append(v)
                    if verbose:
                        print(f"{name} on val_{float(D)}: Validation Loss = {v:.4f}")
                validation_losses[name]["val_avg"].append(float(np.mean(per)))
        if save and num_cycles - cycle - 1 < 5:
            save_results(validation_losses, all_gen_labels, models, results_name, str(num_cycles - cycle), out_dir)
    if save:
        save_results(validation_losses, all_gen_labels, models, results_name, "", out_dir)
    print(datetime.datetime.now())
    return models, validation_losses, all_gen_labels


def save_results(validation_losses, all_gen_labels, models, results_name, path_addition="", out_dir="."):
    """Same dict schema as every reference save_results (e.g. trainModelsPSFNoise.py:14-22)."""
    save_path = f"{out_dir}/training_results_{results_name}{path_addition}.pth"
    torch.save({"validation_losses": validation_losses, "all_labels": all_gen_labels,
                "model_weights": {n: m.state_dict() for n, m in models.items() if m is not None}}, save_path)
    print(f"\nTraining results saved to {save_path}")
    return save_path
