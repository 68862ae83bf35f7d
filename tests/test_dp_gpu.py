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
