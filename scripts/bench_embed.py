"""Micro-benchmark of the two frame-embedding kernels through the C-ABI (hipEvent timing on the launch stream)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from moleculardiffusion_mivit_amd import _native as N

def run(B=2048, T=32, P=64, E=128, iters=20):
    M, K = B * T, P * P
    x = [torch.randn(M, K, device="cuda") for _ in range(2)]
    W = torch.randn(E, K, device="cuda").bfloat16()
    b = torch.randn(E, device="cuda")
    dy = torch.randn(M, E, device="cuda").bfloat16()
    y = torch.empty(M, E, dtype=torch.bfloat16, device="cuda")
    dW = torch.empty(E, K, device="cuda")
    ws = torch.empty(N.lib.mivit_embed_wgrad_bf16_workspace_bytes(M, K, E), dtype=torch.uint8, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    byt = M * K * 4 + E * K * 4 + M * E * 2
    for name, fn in (("fwd", lambda i: N.lib.mivit_embed_fwd_bf16(p(x[i % 2]), p(W), p(b), M, K, E, p(y), st)),
                     ("wgrad", lambda i: N.lib.mivit_embed_wgrad_bf16(p(dy), p(x[i % 2]), M, K, E, p(dW), p(ws), ws.numel(), st))):
        for i in range(3): fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): fn(i)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"{name:6s} B={B} M={M} K={K} E={E}: {ms:.3f} ms  {byt / ms / 1e6:.0f} GB/s (incl. slab reduce for wgrad)")

if __name__ == "__main__":
    for B in (2048, 4096):
        run(B)
