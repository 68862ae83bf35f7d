"""The invariants of the generated gfx950 ISA that the hand-ordered kernels rely on (scripts/isa_check.py): no instruction
touches a register whose inline-asm load is still in flight in embed_fwd_direct2 (and no scratch there), every written-out
LDS-DMA pads the M0 hazard, and the row-store counts that counted waits take as a lower bound are really emitted.
Needs hipcc only (cross-compiles to assembly; no GPU)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_isa_invariants_hold():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_check.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
