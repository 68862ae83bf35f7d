"""Host glue between PyTorch (device memory, streams, autograd, optimizer) and the HIP engine.

`MivitPlan` wraps a native `mivit_plan` (include/mivit_hip.h): it defines the fp32 parameter-arena layout and
runs GeneralTransformer forward / staged backward (reference helpers/models.py:328-361 and its autograd).
`MivitFunction` is the single autograd node the model's forward goes through; with a process group attached it
all-reduces each backward stage's slice of the gradient arena on a side stream while the next stage computes
(data-parallel training, RCCL over xGMI).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch

from . import _native as N

_PRECISIONS = {"fp32": N.F32, "f32": N.F32, "float32": N.F32, "bf16": N.BF16, "bfloat16": N.BF16,
               "fp16": N.F16, "f16": N.F16, "float16": N.F16, "half": N.F16}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream_ptr(device):
    if device.type != "cuda":
        return None
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class MivitPlan:
    """Native plan: parameter-arena layout + forward / backward launch sequences for one model configuration."""

    def __init__(self, *, precision: str, embedding: int, patch_size: int, embed_dim: int, num_heads: int,
                 hidden_dim: int, num_layers: int, activation: int, use_pos_encoding: bool,
                 use_regression_token: bool, fusion: int, global_feature_dim: int, head_hidden: int, output_dim: int):
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}, got {precision!r}")
        self.precision = precision
        self.cfg = N.MivitConfig(N.ABI_VERSION, _PRECISIONS[precision], embedding, patch_size, embed_dim, num_heads,
                                 hidden_dim, num_layers, activation, int(use_pos_encoding), int(use_regression_token),
                                 fusion, int(global_feature_dim or 0), head_hidden, output_dim)
        self._lib = N.lib            # a plan handle belongs to the library instance that created it
        self._h = self._lib.mivit_plan_create(ctypes.byref(self.cfg))
        if not self._h:
            raise N.MivitError(f"mivit_plan_create: {N.last_error()}")
        h = self._h
        n = self._lib.mivit_plan_num_params(h)
        self.param_names: List[str] = [self._lib.mivit_plan_param_name(h, i).decode() for i in range(n)]
        self.param_offsets: List[int] = [self._lib.mivit_plan_param_offset(h, i) for i in range(n)]
        self.param_numels: List[int] = [self._lib.mivit_plan_param_numel(h, i) for i in range(n)]
        self.arena_numel: int = self._lib.mivit_plan_arena_numel(h)
        self.num_stages: int = self._lib.mivit_plan_num_stages(h)
        self.stage_ranges = []
        for s in range(self.num_stages):
            b, e = ctypes.c_int64(), ctypes.c_int64()
            N.check(self._lib.mivit_plan_stage_range(h, s, ctypes.byref(b), ctypes.byref(e)), "stage_range")
            self.stage_ranges.append((b.value, e.value))
        self.embed_dim, self.output_dim = embed_dim, output_dim
        self.embedding = embedding

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        lib = getattr(self, "_lib", None)
        if h and lib is not None:                   # (module may be gone at interpreter exit)
            lib.mivit_plan_destroy(h)

    def workspace_bytes(self, B: int, T: int, need_backward: bool) -> int:
        return self._lib.mivit_plan_workspace_bytes(self._h, B, T, int(need_backward))

    @staticmethod
    def _require_gpu(*tensors):
        for t in tensors:
            if t is not None and t.device.type != "cuda":
                raise RuntimeError("the MiViT HIP path needs GPU tensors (no CPU fallback exists in this package)")

    def forward(self, arena, x, features, B, T, ws, need_backward, out):
        self._require_gpu(arena, x, features)
        N.check(self._lib.mivit_forward(self._h, _ptr(arena), _ptr(x), _ptr(features), B, T, _ptr(ws), ws.numel(),
                                    int(need_backward), _ptr(out), _stream_ptr(x.device)), "mivit_forward")

    def backward(self, arena, x, features, B, T, ws, dout, grads, dfeatures, dx_tokens, s0, s1):
        self._require_gpu(arena, x, dout, grads)
        N.check(self._lib.mivit_backward(self._h, _ptr(arena), _ptr(x), _ptr(features), B, T, _ptr(ws), ws.numel(),
                                     _ptr(dout), _ptr(grads), _ptr(dfeatures), _ptr(dx_tokens), s0, s1,
                                     _stream_ptr(x.device)), "mivit_backward")


class MivitFunction(torch.autograd.Function):
    """out = GeneralTransformer(x, features); inputs after `features` are the arena-backed parameters."""

    @staticmethod
    def forward(ctx, owner, x, features, *params):
        plan: MivitPlan = owner._plan
        arena: torch.Tensor = owner._arena
        x = x.contiguous().float()
        feats = features.contiguous().float() if features is not None else None
        B, T = x.shape[0], x.shape[1]
        need_bwd = any(ctx.needs_input_grad)      # (grad mode is off inside Function.forward; this is the signal)
        ws = torch.empty(plan.workspace_bytes(B, T, need_bwd), dtype=torch.uint8, device=x.device)
        out = torch.empty(B, plan.output_dim, dtype=torch.float32, device=x.device)
        plan.forward(arena, x, feats, B, T, ws, need_bwd, out)
        # x / features go through save_for_backward: autograd then refuses a backward after an in-place write to them.
        # The workspace is private to this node (never visible to the caller) and is released by the first backward.
        ctx.save_for_backward(x, feats if feats is not None else x.new_empty(0))
        ctx.has_feats = feats is not None
        ctx.owner, ctx.ws, ctx.BT = owner, ws, (B, T)
        ctx.arena_version = owner._arena_version
        # every in-place write to a parameter (optimizer.step(), load_state_dict, ...) bumps that parameter's version
        ctx.arena_data_version = sum(p._version for p in owner._arena_params)
        ctx.n_params = len(params)
        ctx.need_x = ctx.needs_input_grad[1]
        ctx.need_f = feats is not None and ctx.needs_input_grad[2]
        return out

    @staticmethod
    def backward(ctx, dout):
        owner = ctx.owner
        plan: MivitPlan = owner._plan
        arena = owner._arena
        if ctx.arena_version != owner._arena_version:
            raise RuntimeError("model parameters were re-allocated between forward and backward")
        if ctx.ws is None:
            raise RuntimeError("backward through the MiViT forward a second time: its saved activations were released by the "
                               "first backward (run the forward again; retain_graph is not supported by this node)")
        if sum(p._version for p in owner._arena_params) != ctx.arena_data_version:
            raise RuntimeError("model parameters were modified in place (optimizer.step()?) between this forward and its "
                               "backward: the backward kernels read the live parameters and would produce wrong gradients")
        B, T = ctx.BT
        x, feats_saved = ctx.saved_tensors
        feats = feats_saved if ctx.has_feats else None
        ws = ctx.ws
        dp = getattr(owner, "_dp", None)            # dp.StagedGradReducer or None
        dout = dout.contiguous().float()
        if dp is not None and dp.world > 1:
            dout = dout / dp.world                  # sum-all-reduce of pre-scaled grads == average
        # Direct parameter gradients (opt-in, GeneralTransformer.direct_param_grads): the gradient arena is a persistent buffer and
        # every parameter's .grad is SET to its cached view of it, instead of handing ~100 views to as many AccumulateGrad nodes
        # (0.5 ms of host time per step: what a launch-bound step at the reference's own batch sizes is made of).
        direct = owner._direct_grads_ready(ctx.needs_input_grad[3:])
        if direct:
            grads = owner._grad_arena_persistent(x.device)
            grads.zero_()
        else:
            grads = torch.zeros(plan.arena_numel, dtype=torch.float32, device=x.device)
        dfeat = torch.empty_like(feats) if ctx.need_f else None
        dx = None
        if plan.embedding == N.EMBED_EXTERNAL and ctx.need_x:
            dx = torch.empty(B, T, plan.embed_dim, dtype=torch.float32, device=x.device)
        if dp is None or dp.world == 1:
            plan.backward(arena, x, feats, B, T, ws, dout, grads, dfeat, dx, 0, plan.num_stages)
        else:
            for s in range(plan.num_stages):
                plan.backward(arena, x, feats, B, T, ws, dout, grads, dfeat, dx, s, s + 1)
                dp.reduce_stage(grads, s)
            dp.finish(grads)
            if dx is not None:
                dx = dx * dp.world                  # d(tokens) feeds the rank-local embedding autograd unscaled
            if dfeat is not None:
                dfeat = dfeat * dp.world
        ctx.ws = None
        if direct:
            for p, v in zip(owner._arena_params, owner._grad_views_cached):
                p.grad = v
            return (None, dx if ctx.need_x else None, dfeat) + (None,) * ctx.n_params
        outs = owner._grad_views(grads)
        return (None, dx if ctx.need_x else None, dfeat) + tuple(outs)
