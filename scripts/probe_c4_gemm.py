"""Config-4 GEMM shapes (E = 512, F = 1024, 2048 x 65 token rows): this package's LDS-DMA GEMMs (mivit_gemm_dma_fwd) beside
PyTorch-ROCm's bf16 matmul (hipBLASLt) on the same operands -- the distance of the hand-written tiles from the vendor library."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from moleculardiffusion_mivit_amd import _native as N
M = 2048 * 65
_p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
_st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for Nn, K in ((1536, 512), (512, 512), (1024, 512), (512, 1024)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(Nn, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(Nn, device="cuda")
    y = torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda")
    fl = 2.0 * M * Nn * K
    t_mine = timeit(lambda: N.check(N.lib.mivit_gemm_dma_fwd(_p(x), K, _p(W), _p(bias), M, Nn, K, 0, None, 0, _p(y), Nn, None, _st()), "gemm"))
    t_lib = timeit(lambda: torch.matmul(x, W.t()))
    t_lib_b = timeit(lambda: torch.nn.functional.linear(x, W, bias.bfloat16()))
    print(f"M={M} N={Nn:5d} K={K:5d}: gemm_dma {t_mine:7.3f} ms {fl / t_mine / 1e9:7.1f} TFLOP/s | hipBLASLt matmul {t_lib:7.3f} ms {fl / t_lib / 1e9:7.1f} | + bias {t_lib_b:7.3f} ms {fl / t_lib_b / 1e9:7.1f}", flush=True)
