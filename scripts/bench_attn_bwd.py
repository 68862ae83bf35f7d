"""Micro-benchmark of the attention core's backward (mivit_attention_bwd: qkv, dctx -> dqkv) at the headline shape.
python scripts/bench_attn_bwd.py [B=16384] [S=33] [H=4] [Dh=32]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from moleculardiffusion_mivit_amd import _native as N
B, S, H, Dh = [int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 16384), (2, 33), (3, 4), (4, 32))]
E = H * Dh
qkv = torch.randn(B, S, 3 * E, device="cuda").bfloat16()
dctx = torch.randn(B, S, E, device="cuda").bfloat16()
dqkv = torch.empty_like(qkv)
_p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
fn = lambda: N.check(N.lib.mivit_attention_bwd(N.BF16, _p(qkv), _p(dctx), B, S, H, Dh, _p(dqkv), st), "attention_bwd")
for _ in range(3): fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): fn()
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 20 * 1e3
byt = B * S * E * 2 * 7
print(f"attention_bwd B={B} S={S} H={H} Dh={Dh}: {us:8.1f} us   {byt / us / 1e3:7.1f} GB/s  checksum {float(dqkv.float().abs().sum()):.6e}")
