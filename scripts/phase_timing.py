"""Phase-level cycle breakdown of mlp_block_bwd (and whatever else carries PT_MARK): builds csrc/fused_bwd.hip with
-DMIVIT_PHASE_TIMING into libmivit_hip_timing.so (never shipped, never loaded by the product), runs the kernel at the bench
shape and prints s_memtime cycles per tile and phase.
    python scripts/phase_timing.py [B]"""
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moleculardiffusion_mivit_amd.csrc import build as b  # noqa: E402

b.build(verbose=False)
obj = os.path.join(b.OBJDIR, "fused_bwd.timing.o")
lib = os.path.join(b.PKG, "libmivit_hip_timing.so")
subprocess.check_call([b._hipcc()] + b.FLAGS + ["-DMIVIT_PHASE_TIMING", "-c", os.path.join(b.HERE, "fused_bwd.hip"), "-o", obj],
                      stderr=subprocess.DEVNULL)
objs = [os.path.join(b.OBJDIR, s.replace(".hip", ".o")) if s != "fused_bwd.hip" else obj for s in b.SOURCES]
objs += [os.path.join(b.OBJDIR, s.replace(".hip", ".f16.o")) for s in b.ELEM_SOURCES]          # the IEEE-half builds of the streaming units
objs += [os.path.join(b.OBJDIR, s.replace(".hip", ext)) for s in b.WIDTH_SOURCES for ext in (".w64.o", ".w64.f16.o")]   # ... and the width-64 builds
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)

import torch  # noqa: E402
from moleculardiffusion_mivit_amd import _native as N, ops  # noqa: E402

N.lib = N._load(lib)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
S, E, FH = 33, 128, 256
M = B * S
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc  # noqa: E731
n_in, dy = rn(M, E).bfloat16(), rn(M, E).bfloat16()
gi, bi, rstd = 1 + 0.1 * rn(E), 0.1 * rn(E), 1 + 0.1 * rn(M).abs()
W1, b1 = rn(FH, E, sc=1 / math.sqrt(E)).bfloat16(), 0.1 * rn(FH)
W2 = rn(E, FH, sc=1 / math.sqrt(FH)).bfloat16()
for _ in range(2):
    ops.mlp_block_bwd(dy, n_in, rstd, gi, n_in, gi, bi, W1, b1, W2)
torch.cuda.synchronize()
