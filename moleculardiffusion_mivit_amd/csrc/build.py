"""Build libmivit_hip.so (gfx950) in-tree with hipcc.  No torch dependency: the library is plain C-ABI.

    python -m moleculardiffusion_mivit_amd.csrc.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
SOURCES = ["gemm.hip", "norm.hip", "attention.hip", "attention_fast.hip", "embed.hip", "rowstream.hip", "wavestream.hip", "gemm_dma.hip", "wgrad_dma.hip", "wgrad_small.hip", "fused_fwd.hip", "fused_bwd.hip", "render.hip", "deepresnet.hip", "deepresnet_train.hip", "misc.hip", "engine.hip"]
HEADERS = [os.path.join(HERE, "common.h"), os.path.join(HERE, "stream_prims.h"), os.path.join(HERE, "elem.h"),
           os.path.join(ROOT, "include", "mivit_hip.h")]
# the streaming kernels whose element type is chosen per translation unit (elem.h): each is compiled a second time with
# -DMIVIT_ELEM_F16 (IEEE half instead of bf16, every external suffixed _f16) and both objects go into the library
ELEM_SOURCES = ["rowstream.hip", "wavestream.hip", "wgrad_dma.hip", "wgrad_small.hip", "attention_fast.hip", "embed.hip",
                "fused_fwd.hip", "fused_bwd.hip"]
# the fused encoder-layer blocks are in addition compiled per layer width (elem.h): -DMIVIT_WIDTH64 = the reference's shipped
# E = 64 / F = 128 / head dim 16, externals suffixed _w64 (and _w64_f16)
WIDTH_SOURCES = ["fused_fwd.hip", "fused_bwd.hip"]
LIB = os.path.join(PKG, "libmivit_hip.so")
# the same library with every counted s_waitcnt vmcnt(N) of stream_prims.h::wait_vm turned into vmcnt(0) (-DMIVIT_STRICT_WAITS).
# TEST INFRASTRUCTURE ONLY: tests/test_strict_waits_gpu.py runs the bench-scale shapes through both and requires bitwise-equal
# results; the product (_native.py) never loads it.  Only the sources that call wait_vm are compiled twice.
LIB_STRICT = os.path.join(PKG, "libmivit_hip_strict.so")
WAIT_SOURCES = ["embed.hip", "rowstream.hip", "gemm_dma.hip", "wgrad_dma.hip", "fused_bwd.hip"]
OBJDIR = os.path.join(HERE, "build")
# -amdgpu-mfma-vgpr-form: MFMA results in arch VGPRs where they fit.  By default the accumulators go to AGPRs and every value a
# VALU instruction consumes afterwards (softmax, bias, packing, stores) costs a v_accvgpr_read first: attention_fast.hip 23 k
# -> 5 k such moves, attn_block_fwd 1149 -> 238 per sequence (470 -> 358 registers), the scratch spills of mlp_block_fwd gone,
# wgrad / gemm kernels 144-252 -> 94-186 registers.  Kernels that need more than 256 accumulators + operands (mlp_block_bwd,
# the 128 x 128 wave tile) still get AGPRs for the part that does not fit.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-inline-asm", "-fno-gpu-rdc", "-mllvm", "-amdgpu-mfma-vgpr-form"]


# experiments: extra compiler flags for every object (e.g. MIVIT_EXTRA_HIPCC_FLAGS="-DMIVIT_NO_ASM_PF" python build.py --force)
FLAGS += os.environ.get("MIVIT_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, strict=True):
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s, os.path.abspath(__file__)] + HEADERS):        # the flags live in this file
            jobs.append((s, o, []))
        variants = [(".strict.o", ["-DMIVIT_STRICT_WAITS"])] if strict and src in WAIT_SOURCES else []
        if src in ELEM_SOURCES:
            variants.append((".f16.o", ["-DMIVIT_ELEM_F16"]))
            if strict and src in WAIT_SOURCES:
                variants.append((".f16.strict.o", ["-DMIVIT_ELEM_F16", "-DMIVIT_STRICT_WAITS"]))
        if src in WIDTH_SOURCES:
            for ext, extra in list(variants) + [(".o", [])]:
                variants.append((".w64" + ext, ["-DMIVIT_WIDTH64"] + extra))
        for ext, extra in variants:
            o = os.path.join(OBJDIR, src.replace(".hip", ext))
            if force or _stale(o, [s, os.path.abspath(__file__)] + HEADERS):
                jobs.append((s, o, extra))

    def compile_one(job):
        s, o, extra = job
        cmd = [hipcc] + FLAGS + extra + ["-c", s, "-o", o]
        if verbose:
            print("[mivit build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJDIR, s.replace(".hip", ".o")) for s in SOURCES]
    objs += [os.path.join(OBJDIR, s.replace(".hip", ".f16.o")) for s in ELEM_SOURCES]
    objs += [os.path.join(OBJDIR, s.replace(".hip", ext)) for s in WIDTH_SOURCES for ext in (".w64.o", ".w64.f16.o")]
    sobjs = [os.path.join(OBJDIR, s.replace(".hip", ".strict.o" if s in WAIT_SOURCES else ".o")) for s in SOURCES]
    sobjs += [os.path.join(OBJDIR, s.replace(".hip", ".f16.strict.o" if s in WAIT_SOURCES else ".f16.o")) for s in ELEM_SOURCES]
    sobjs += [os.path.join(OBJDIR, s.replace(".hip", ext)) for s in WIDTH_SOURCES
              for ext in ((".w64.strict.o", ".w64.f16.strict.o") if strict and s in WAIT_SOURCES else (".w64.o", ".w64.f16.o"))]
    for lib, ob in ((LIB, objs), (LIB_STRICT, sobjs)) if strict else ((LIB, objs),):
        if force or jobs or _stale(lib, ob):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + ob
            if verbose:
                print("[mivit build]", " ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
