"""GPU: the data-parallel path on real kernels.  Two ranks share the one GPU of the test box (gloo backend: RCCL
refuses two ranks on one device), so this exercises the staged native backward, the per-stage all-reduce on the
communication stream and the stream/event ordering; the RCCL transport itself is what `bench.py --gpus N` uses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from oracle import mivit_oracle as orc
from util import build_product_model

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from moleculardiffusion_mivit_amd import dp
        torch.cuda.set_device(0)
        cfg = orc.MiViTConfig(embedding="linear", patch_size=16, embed_dim=64, num_heads=4, hidden_dim=128, num_layers=3,
                              use_pos_encoding=True)
        params = orc.closed_form_params(cfg)
        B = 16
        x, y, _ = orc.closed_form_batch(B, 12, 16, salt=1)
        for prec, tol in (("fp32", 2e-5), ("bf16", 2e-5)):
            torch.manual_seed(rank)
            single = build_product_model(cfg, prec, params)
            out = single(x.cuda())
            F.mse_loss(out, y.cuda()).backward()
            ref = {k: p.grad.clone() for k, p in single.named_parameters()}
            model = build_product_model(cfg, prec, None if rank else params)   # rank 1 starts from random weights
            dp.attach(model)                                                  # ... and receives rank 0's
            sh = slice(rank * B // world, (rank + 1) * B // world)
            F.mse_loss(model(x[sh].cuda()), y[sh].cuda()).backward()
            torch.cuda.synchronize()
            gscale = max(float(g.abs().max()) for g in ref.values())
            worst = max(float((p.grad - ref[k]).abs().max()) / (float(ref[k].abs().max()) + 1e-3 * gscale)
                        for k, p in model.named_parameters())
            # fp32: shard-average == full batch up to summation order.  bf16: per-shard activations are rounded the
            # same way as in the full batch (row-independent kernels), so the same bound holds.
            assert worst < (tol if prec == "fp32" else 3e-3), (prec, worst)
        ret[rank] = "ok"
    except Exception:  # noqa: BLE001
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
        raise
    finally:
        dist.destroy_process_group()


def test_data_parallel_two_ranks_one_gpu():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


# ---- synchronised BatchNorm for the DeepResNet embedding ---------------------------------------
def _sync_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import copy
        from moleculardiffusion_mivit_amd import dp
        from moleculardiffusion_mivit_amd.helpers.models import DeepResNetEmbedding
        torch.cuda.set_device(0)

        # The single-device references below run through the SAME staged kernels on a group of one rank, and every rank's
        # frame count is a multiple of 12 (12-frame sequences): the workgroups then hold the same frames in both runs, their
        # fp32 partial sums are identical, the statistics agree to fp64 rounding and so does every ReLU mask.  (With other
        # groupings, or against the un-staged path, the statistics differ in the last fp32 bit, one pre-activation within
        # rounding of zero flips its mask every few runs and shows up at 1e-3 -- DESIGN.md section 5a.)
        solo = [dist.new_group([r]) for r in range(world)][rank]

        def worst(got, ref):
            gscale = max(float(g.abs().max()) for g in ref.values())
            return max(float((got[k] - ref[k]).abs().max()) / (float(ref[k].abs().max()) + 1e-3 * gscale) for k in ref)

        # (1) the embedding alone, UNEVEN shards (3 + 2 sequences; 13-pixel fp32 frames are cut into tiles): a sum loss,
        # so the job's gradient is the plain sum of the ranks' gradients
        for prec, P, T, tol in (("fp32", 9, 12, 2e-4), ("fp32", 13, 12, 2e-4), ("bf16", 9, 12, 3e-2)):
            torch.manual_seed(7)                                   # same module and data on both ranks
            full = DeepResNetEmbedding(P, 64)
            with torch.no_grad():
                for m in full.modules():
                    if isinstance(m, torch.nn.BatchNorm2d):
                        m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
                        m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 2.0)
            full = full.cuda().train()
            full.__dict__["_mivit_precision"] = prec
            shard = copy.deepcopy(full).sync_batchnorm()
            full.sync_batchnorm(solo)
            shard.__dict__["_mivit_precision"] = prec
            x = torch.rand(5, T, P, P, device="cuda") * 1.5 - 0.25
            wgt = torch.randn(5, T, 64, device="cuda")
            out_full = full(x)                                     # native kernels, whole minibatch on one device
            (out_full * wgt).sum().backward()
            sh = slice(0, 3) if rank == 0 else slice(3, 5)
            out = shard(x[sh])
            (out * wgt[sh]).sum().backward()
            torch.cuda.synchronize()
            assert float((out.detach() - out_full.detach()[sh]).abs().max()) / float(out_full.detach().abs().max()) < tol, prec
            got = {}
            for k, p in shard.named_parameters():
                g = p.grad.clone()
                dist.all_reduce(g)
                got[k] = g
            ref = {k: p.grad for k, p in full.named_parameters()}
            assert worst(got, ref) < tol, (prec, P, worst(got, ref))
            for (k, a), (_, b) in zip(shard.named_buffers(), full.named_buffers()):   # running statistics: job-wide
                assert float((a.float() - b.float()).abs().max()) <= tol * (float(b.float().abs().max()) + 1e-6), (prec, k)

        # (2) the whole model through dp.attach(sync_batchnorm=True): mean loss over equal shards == full-batch step
        cfg = orc.MiViTConfig(embedding="deepresnet", patch_size=9, embed_dim=64, num_heads=4, hidden_dim=128, num_layers=2)
        params = orc.closed_form_params(cfg)
        B = 8
        x, y, _ = orc.closed_form_batch(B, 12, 9, salt=3)
        single = build_product_model(cfg, "fp32", params).train()
        single.embedding.sync_batchnorm(solo)
        F.mse_loss(single(x.cuda()), y.cuda()).backward()
        ref = {k: p.grad.clone() for k, p in single.named_parameters()}
        model = build_product_model(cfg, "fp32", None if rank else params).train()
        dp.attach(model, sync_batchnorm=True)
        sh = slice(rank * B // world, (rank + 1) * B // world)
        F.mse_loss(model(x[sh].cuda()), y[sh].cuda()).backward()
        dp.finish_external_grads(model)
        torch.cuda.synchronize()
        got = {k: p.grad for k, p in model.named_parameters()}
        assert worst(got, ref) < 2e-4, worst(got, ref)
        for (k, a), (_, b) in zip(model.named_buffers(), single.named_buffers()):
            assert float((a.float() - b.float()).abs().max()) <= 2e-4 * (float(b.float().abs().max()) + 1e-6), k
        ret[rank] = "ok"
    except Exception:  # noqa: BLE001
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
        raise
    finally:
        dist.destroy_process_group()


def test_deepresnet_synchronised_batchnorm_two_ranks_one_gpu():
    """Two ranks, each with a shard of the minibatch, must reproduce the single-device step on the whole minibatch:
    tokens, every parameter gradient and the running statistics (fp32 2e-4; bf16 activations 3e-2)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sync_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


# ---- the mirrored training loop under data parallelism ----------------------------------------------------------------
def _loop_worker(rank, world, port, ret, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    kw = dict(num_cycles=2, N=3, seed=0, embedding="deepresnet", psf_indices=[1], noise_indices=[0], include_resnet=False,
              TrainingDs_list=([1, 1], [5, 1], [9, 1]))                # 9 sequences per cycle: one minibatch, split 4 + 5
    try:
        from moleculardiffusion_mivit_amd.experiments.PSFNoise import trainModelsPSFNoise as TM
        from moleculardiffusion_mivit_amd.experiments.PSFNoise import trainSettingsPSFNoise as S
        S.adaptive_batch_size = -1                                     # fixed minibatch of 16 (the default doubles from 1)
        torch.manual_seed(0)
        single, vl_single, _ = TM.run_training(out_dir=os.path.join(tmp, f"single{rank}"), save=False, **kw)
        ref = {k: v.detach().clone() for k, v in single["tr_1_0"].state_dict().items()}
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.manual_seed(100 + rank)                                  # different initial weights: rank 0's are broadcast...
        os.makedirs(os.path.join(tmp, "dp"), exist_ok=True)
        if rank == 0:
            torch.manual_seed(0)                                       # ... and rank 0 starts where the single run started
        models, vl, _ = TM.run_training(out_dir=os.path.join(tmp, "dp"), save=True, **kw)
        got = models["tr_1_0"].state_dict()
        for k, v in ref.items():
            if v.dtype.is_floating_point:
                err = float((got[k] - v).abs().max()) / (float(v.abs().max()) + 1e-6)
                assert err < 2e-3, (k, err)
        assert abs(vl["tr_1_0"]["val_avg"][-1] - vl_single["tr_1_0"]["val_avg"][-1]) < 2e-2 * abs(vl_single["tr_1_0"]["val_avg"][-1])
        dist.barrier()
        files = sorted(os.listdir(os.path.join(tmp, "dp")))
        assert files == ["training_results_PSFNoise.pth", "training_results_PSFNoise1.pth", "training_results_PSFNoise2.pth"], files
        ret[rank] = "ok"
    except Exception:  # noqa: BLE001
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
        raise
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_training_loop_two_ranks_matches_single_process(tmp_path):
    """experiments/PSFNoise run_training under torch.distributed (2 ranks, DeepResNet embedding -> synchronised BatchNorm,
    uneven 4 + 5 split of the 9-sequence minibatch) ends with the weights and validation loss of the single-process run;
    only rank 0 writes checkpoints."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    for r in range(world):
        os.makedirs(tmp_path / f"single{r}", exist_ok=True)
    mp.spawn(_loop_worker, args=(world, _free_port(), ret, str(tmp_path)), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def _rccl_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world)          # backend "nccl" IS RCCL on ROCm
        from moleculardiffusion_mivit_amd import dp
        cfg = orc.MiViTConfig(embedding="linear", patch_size=16, embed_dim=64, num_heads=4, hidden_dim=128, num_layers=2)
        params = orc.closed_form_params(cfg)
        x, y, _ = orc.closed_form_batch(8, 12, 16, salt=1)
        single = build_product_model(cfg, "bf16", params)
        F.mse_loss(single(x.cuda()), y.cuda()).backward()
        ref = {k: p.grad.clone() for k, p in single.named_parameters()}
        model = build_product_model(cfg, "bf16", params)
        dp.attach(model)                                  # broadcast of the arena over RCCL
        # a one-rank job skips the collectives (world == 1); force the staged path so that RCCL's all-reduce kernels really
        # run on the communication stream behind the per-stage events (sum over one rank = identity; 1 / world = 1 / 2 here)
        model._dp.world = 2
        F.mse_loss(model(x.cuda()), y.cuda()).backward()
        torch.cuda.synchronize()
        for k, p in model.named_parameters():
            assert torch.equal(p.grad * 2, ref[k]) or float((p.grad * 2 - ref[k]).abs().max()) <= 1e-6 * float(ref[k].abs().max() + 1e-12), k
        t = torch.arange(1024, device="cuda", dtype=torch.float32)
        dist.all_reduce(t)
        dist.broadcast(t, src=0)
        torch.cuda.synchronize()
        assert float(t.sum()) == 1023 * 1024 / 2
        ret[rank] = "ok"
    except Exception:  # noqa: BLE001
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
        raise
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_rccl_backend_single_rank_staged_reduce():
    """RCCL itself on this hardware (one rank: what a one-GPU box allows): communicator creation, broadcast of the parameter
    arena, and the staged per-slice all-reduces issued on the communication stream behind the backward's events."""
    ret = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
    assert dict(ret) == {0: "ok"}, dict(ret)
