"""Data-parallel training support: one process per GPU, gradient averaging with RCCL over xGMI.

The model's backward runs in stages (head, encoder layers L-1..0, embedding); the gradients of a stage are a
contiguous slice of the fp32 gradient arena, final as soon as that stage's kernels are enqueued.  Each slice is
all-reduced on a side stream while the next stage computes, so communication hides under backward; there is no
copy into buckets (the arena IS the bucket) and no extra collective on the data path.  Samples are sharded
contiguously over ranks; per-rank losses are means over equal shards, so sum(all-reduce) of grads pre-scaled by
1/world equals the reference's full-batch MSELoss(mean) gradient.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class StagedGradReducer:
    """All-reduce (sum) of arena slices, stage by stage.  GPU tensors: asynchronous on a communication stream
    ordered after the producing kernels by an event.  CPU tensors (gloo tests): synchronous."""

    def __init__(self, stage_ranges, group=None):
        self.stage_ranges = list(stage_ranges)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._comm = None
        self._events = {}
        self._done = {}
        self._last_done = None

    def reduce_stage(self, grads: torch.Tensor, stage: int):
        b, e = self.stage_ranges[stage]
        if e <= b or self.world == 1:
            return
        if grads.device.type == "cuda":
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=grads.device)
            main = torch.cuda.current_stream(grads.device)
            ev = self._events.get(stage)
            if ev is None:
                ev = self._events[stage] = torch.cuda.Event()
            ev.record(main)
            self._comm.wait_event(ev)
            with torch.cuda.stream(self._comm):
                dist.all_reduce(grads[b:e], group=self.group)
                done = self._done.get(stage)
                if done is None:
                    done = self._done[stage] = torch.cuda.Event()
                done.record(self._comm)
                self._last_done = done
        else:
            dist.all_reduce(grads[b:e], group=self.group)

    def finish(self, grads: torch.Tensor):
        """Order whatever consumes the gradients (the optimizer step) after the LAST slice's all-reduce -- an event wait, not a
        stream join: collectives of other models or of `finish_external_grads` queued on the communication stream later do
        not hold the main stream back.  The last slice is the frame embedding's (its weight gradient is the last kernel of
        backpropagation: nothing is left to overlap its 2.1 MB all-reduce with -- the one exposed collective of a step)."""
        if grads.device.type == "cuda" and self._comm is not None and self._last_done is not None:
            torch.cuda.current_stream(grads.device).wait_event(self._last_done)
            grads.record_stream(self._comm)
            self._last_done = None


def broadcast_parameters(model, src: int = 0, group=None):
    """Identical initialisation on every rank: one broadcast of the whole parameter arena (+ any tensors that
    live outside it, e.g. a DeepResNet embedding and its BatchNorm buffers)."""
    arena = getattr(model, "_arena", None)
    inside = set()
    if arena is not None:
        dist.broadcast(arena, src=src, group=group)
        inside = {id(p) for p in model._arena_params}
    for p in model.parameters():
        if id(p) not in inside:
            dist.broadcast(p.data, src=src, group=group)
    for b in model.buffers():
        dist.broadcast(b, src=src, group=group)


def attach(model, group=None, broadcast: bool = True, sync_batchnorm=None):
    """Make `model` (a GeneralTransformer) data-parallel over `group`: gradients produced by its backward are
    averaged across ranks, overlapped stage by stage.  Parameters outside the arena (external embeddings) are
    averaged by `finish_external_grads` after backward.  `sync_batchnorm`: a DeepResNet embedding normalises with
    the statistics of the whole job's minibatch instead of the rank's shard -- what the
    reference's single device computes, hence the default (None = on whenever the embedding has BatchNorm; False
    keeps per-rank statistics, as stock DDP does; True insists and raises for embeddings without BatchNorm)."""
    if broadcast:
        broadcast_parameters(model, 0, group)
    emb = getattr(model, "embedding", None)
    if sync_batchnorm is None:
        sync_batchnorm = hasattr(emb, "sync_batchnorm")
    if sync_batchnorm:
        if not hasattr(emb, "sync_batchnorm"):
            raise ValueError("sync_batchnorm=True needs a DeepResNetEmbedding (the only embedding with BatchNorm)")
        emb.sync_batchnorm(group)
    model._dp = StagedGradReducer(model._plan.stage_ranges, group)
    # Models on the composed path (dropout > 0, a free-form activation, a non-ReLU head: GeneralTransformer._forward_composed)
    # differentiate through ordinary autograd, never through MivitFunction.backward -- the staged reducer above is not
    # reached.  Their gradients are averaged by a hook per parameter instead, fired when autograd has accumulated that
    # parameter's gradient (a compatibility path: ~50 small collectives per step, correct before fast).
    for h in getattr(model, "_dp_hooks", []):
        h.remove()
    model._dp_hooks = []
    if getattr(model, "_composed", False) and model._dp.world > 1:
        world = model._dp.world

        def _average(p, _world=world, _group=group):
            p.grad.mul_(1.0 / _world)
            dist.all_reduce(p.grad, group=_group)

        model._dp_hooks = [p.register_post_accumulate_grad_hook(_average) for p in model.parameters() if p.requires_grad]
    return model


def finish_external_grads(model, group=None):
    """Average the gradients of parameters that are not in the arena (a DeepResNet embedding's 23 tensors; call after
    loss.backward()): ONE collective on one flat buffer, pre-scaled by 1/world, issued on the communication stream so
    it runs beside whatever the main stream still has queued (the optimizer waits for it through the stream join)."""
    world = dist.get_world_size(group)
    if world == 1 or getattr(model, "_dp_hooks", None):      # composed-path models: every parameter was averaged by its hook
        return
    inside = {id(p) for p in getattr(model, "_arena_params", [])}
    ext = [p for p in model.parameters() if id(p) not in inside and p.grad is not None]
    if not ext:
        return
    dev = ext[0].grad.device
    reducer = getattr(model, "_dp", None)
    if dev.type == "cuda":
        comm = reducer._comm if reducer is not None and reducer._comm is not None else torch.cuda.Stream(device=dev)
        if reducer is not None:
            reducer._comm = comm
        main = torch.cuda.current_stream(dev)
        comm.wait_stream(main)
        with torch.cuda.stream(comm):
            flat = torch.cat([p.grad.reshape(-1) for p in ext]).mul_(1.0 / world)
            dist.all_reduce(flat, group=group)
            off = 0
            for p in ext:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        main.wait_stream(comm)
        for p in ext:
            p.grad.record_stream(comm)
    else:
        flat = torch.cat([p.grad.reshape(-1) for p in ext]).mul_(1.0 / world)
        dist.all_reduce(flat, group=group)
        off = 0
        for p in ext:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
