"""The wide-layer forward GEMMs of config 4 (E=512, F=1024, B=2048 x 65 tokens) through the C-ABI, a few launches each:
the workload behind `rocprofv3 --pmc ...` / `--kernel-trace --stats` for csrc/gemm_dma.hip (MIVIT_GEMM_DMA_VARIANT picks the tile)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from moleculardiffusion_mivit_amd import _native as N

M = 2048 * 65
p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for (n, k) in ((1536, 512), (512, 512), (1024, 512), (512, 1024)):
    x = torch.randn(M, k, device="cuda").bfloat16(); W = torch.randn(n, k, device="cuda").bfloat16()
    b = torch.randn(n, device="cuda"); y = torch.empty(M, n, dtype=torch.bfloat16, device="cuda")
    dy = torch.randn(M, n, device="cuda").bfloat16(); dx = torch.empty(M, k, dtype=torch.bfloat16, device="cuda")
    for _ in range(2):
        N.check(N.lib.mivit_gemm_dma_fwd(p(x), k, p(W), p(b), M, n, k, 0, None, n, p(y), n, None, st), "fwd")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        N.check(N.lib.mivit_gemm_dma_fwd(p(x), k, p(W), p(b), M, n, k, 0, None, n, p(y), n, None, st), "fwd")
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    e0.record()
    for _ in range(reps):
        N.check(N.lib.mivit_gemm_dma_dgrad(p(dy), n, p(W), M, n, k, 0, None, k, None, k, p(dx), k, st), "dgrad")
    e1.record(); torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / reps
    fl = 2.0 * M * n * k
    print(f"N={n:5d} K={k:5d}: fwd {ms:7.3f} ms {fl / ms / 1e9:7.1f} TFLOP/s   dgrad {ms2:7.3f} ms {fl / ms2 / 1e9:7.1f} TFLOP/s", flush=True)
