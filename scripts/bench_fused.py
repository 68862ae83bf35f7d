"""Micro-benchmark of the fused encoder-layer block kernels at the headline shape (B sequences of 33 tokens, E=128) or, with
E = 64, the reference's shipped width (Framerate shape: 31 tokens).
    python scripts/bench_fused.py [B] [E] [S]
Prints microseconds per launch, achieved HBM GB/s on the algorithmic bytes and MFMA TFLOP/s on the algorithmic FLOPs."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculardiffusion_mivit_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
E = int(sys.argv[2]) if len(sys.argv) > 2 else 128
S = int(sys.argv[3]) if len(sys.argv) > 3 else (33 if E == 128 else 31)
FH = 2 * E
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
n_in = rn(B, S, E).bfloat16()
gi, bi = 1 + 0.1 * rn(E), 0.1 * rn(E)
Wqkv, bqkv = rn(3 * E, E, sc=1 / math.sqrt(E)).bfloat16(), 0.1 * rn(3 * E)
Wo, bo = rn(E, E, sc=1 / math.sqrt(E)).bfloat16(), 0.1 * rn(E)
W1, b1 = rn(FH, E, sc=1 / math.sqrt(E)).bfloat16(), 0.1 * rn(FH)
W2, b2 = rn(E, FH, sc=1 / math.sqrt(FH)).bfloat16(), 0.1 * rn(E)
M = B * S


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


U = M * E * 2
import ctypes
from moleculardiffusion_mivit_amd import _native as N
_ab = dict(ctx=torch.empty(B, S, E, dtype=torch.bfloat16, device=dev), n=torch.empty(B, S, E, dtype=torch.bfloat16, device=dev),
           rstd=torch.empty(B, S, device=dev), qkv=torch.empty(B, S, 3 * E, dtype=torch.bfloat16, device=dev))
_p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
_bq, _bo = bqkv.float(), bo.float()


def attn_train():       # what the engine launches in training: ctx, n, rstd + the q|k|v store, nothing else
    N.check(ops._fused_entry("mivit_attn_block_fwd", E)(_p(n_in), _p(gi), _p(bi), _p(Wqkv), _p(_bq), _p(Wo), _p(_bo), _p(gi), _p(bi), B, S, _p(_ab["ctx"]),
                                       _p(_ab["n"]), _p(_ab["rstd"]), None, None, None, _p(_ab["qkv"]),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "attn_block_fwd")


cases = {
    "attn_block_fwd lean": (lambda: ops.attn_block_fwd(n_in, gi, bi, Wqkv, bqkv, Wo, bo, gi, bi), 3 * U,
                            M * (8 * E * E + 4 * S * E)),
    "attn_block_fwd +legacy outputs": (lambda: ops.attn_block_fwd(n_in, gi, bi, Wqkv, bqkv, Wo, bo, gi, bi, extras=True), 8 * U,
                                       M * (8 * E * E + 4 * S * E)),
    "attn_block_fwd + q|k|v (training)": (lambda: attn_train(), 6 * U, M * (8 * E * E + 4 * S * E)),
    "mlp_block_fwd lean": (lambda: ops.mlp_block_fwd(n_in.view(M, E), gi, bi, W1, b1, W2, b2, gi, bi), 2 * U, M * 4 * E * FH),
    "mlp_block_fwd +legacy outputs": (lambda: ops.mlp_block_fwd(n_in.view(M, E), gi, bi, W1, b1, W2, b2, gi, bi, extras=True),
                                      8 * U, M * 4 * E * FH),
}
if hasattr(ops, "mlp_block_bwd"):
    dy = rn(M, E).bfloat16()
    rstd = 1 + 0.1 * rn(M).abs()
    cases["mlp_block_bwd"] = (lambda: ops.mlp_block_bwd(dy, n_in.view(M, E), rstd, gi, n_in.view(M, E), gi, bi, W1, b1, W2), 4 * U,
                              M * 10 * E * FH)
if hasattr(ops, "attn_out_bwd"):
    cases["attn_out_bwd"] = (lambda: ops.attn_out_bwd(dy, n_in.view(M, E), rstd, gi, n_in.view(M, E), Wo), 5 * U, M * 4 * E * E)
if hasattr(ops, "qkv_bwd"):
    dqkv = rn(M, 3 * E).bfloat16()
    cases["qkv_bwd"] = (lambda: ops.qkv_bwd(dqkv, n_in.view(M, E), Wqkv, dy), 6 * U, M * 12 * E * E)
for name, (fn, byt, fl) in cases.items():
    us = timeit(fn)
    print(f"{name:34s} {us:9.1f} us   {byt / us / 1e3:8.1f} GB/s   {fl / us / 1e6:8.1f} TFLOP/s", flush=True)
