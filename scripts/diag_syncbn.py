"""Diagnostic: 2 gloo ranks on one GPU; per-parameter gradient error of (a) the native full-minibatch step and (b) the
synchronised-BatchNorm sharded step against the PyTorch-ROCm conv stack on the whole minibatch."""
import copy
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, P, prec, T):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from moleculardiffusion_mivit_amd.helpers.models import DeepResNetEmbedding
    torch.cuda.set_device(0)
    torch.manual_seed(7)
    full = DeepResNetEmbedding(P, 64).cuda().train()
    full.__dict__["_mivit_precision"] = prec
    tor = copy.deepcopy(full)
    shard = copy.deepcopy(full).sync_batchnorm()
    x = torch.rand(5, T, P, P, device="cuda") * 1.5 - 0.25
    wgt = torch.randn(5, T, 64, device="cuda")
    (full(x) * wgt).sum().backward()
    os.environ["MIVIT_NO_DEEPRESNET_TRAIN"] = "1"
    (tor(x) * wgt).sum().backward()
    os.environ.pop("MIVIT_NO_DEEPRESNET_TRAIN")
    sh = slice(0, 3) if rank == 0 else slice(3, 5)
    (shard(x[sh]) * wgt[sh]).sum().backward()
    torch.cuda.synchronize()
    ref = {k: p.grad for k, p in tor.named_parameters()}
    gscale = max(float(g.abs().max()) for g in ref.values())
    for (k, pf), (_, ps) in zip(full.named_parameters(), shard.named_parameters()):
        g = ps.grad.clone(); dist.all_reduce(g)
        den = float(ref[k].abs().max()) + 1e-3 * gscale
        if rank == 0:
            print(f"{k:32s} full {float((pf.grad - ref[k]).abs().max()) / den:.2e}  sync {float((g - ref[k]).abs().max()) / den:.2e}")
    dist.destroy_process_group()


if __name__ == "__main__":
    P = int(sys.argv[1]); prec = sys.argv[2]; T = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    mp.spawn(worker, args=(2, 29513, P, prec, T), nprocs=2, join=True)
