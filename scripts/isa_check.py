"""Build-time checks of the generated gfx950 ISA for the invariants the hand-ordered kernels rely on (csrc/stream_prims.h).

Run by `__graft_entry__.build()` and by tests/test_isa_invariants.py (CPU, needs hipcc only):

  1. embed_fwd_direct2 (csrc/embed.hip) loads its frame fragments with inline-asm `global_load_dwordx4` whose destination
     registers the compiler believes are defined immediately.  Between such a load and the counted `s_waitcnt vmcnt(N)` that
     covers it NO instruction may read or write those registers (no v_mov, no spill, no re-use as an address), and the
     kernel must not use scratch.  The check replays the kernel's instruction stream (prologue, then the loop twice) with an
     in-order queue of outstanding VM operations.
  2. Every written-out LDS-DMA sets M0 and pads one wait state before the DMA instruction (`s_mov_b32 m0` / `s_nop` /
     `global_load_lds_*`), in every kernel of every source that includes stream_prims.h.
  3. The store counts that `wait_vm<N>` call sites count on as a LOWER bound: mlp_block_bwd / attn_out_bwd issue (at least) four
     8-byte row stores per tile and wave; a rowstream epilogue pass issues at least one 16-byte store.
  4. No inline-asm VALU instruction is the first reader of an MFMA result (no hazard wait states would separate them).

Exit status 0 = all invariants hold; otherwise the violations are listed and the status is 1.
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "moleculardiffusion_mivit_amd", "csrc")
sys.path.insert(0, ROOT)


def device_asm(src, extra=()):
    from moleculardiffusion_mivit_amd.csrc import build as b
    out = os.path.join(tempfile.gettempdir(), "mivit_isa_" + os.path.basename(src).replace(".hip", "".join(extra).replace("-D", "_") + ".s"))
    srcp = os.path.join(CSRC, src)
    deps = [srcp, os.path.join(CSRC, "common.h"), os.path.join(CSRC, "stream_prims.h"), os.path.join(CSRC, "elem.h"), b.__file__]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        cmd = [b._hipcc()] + b.FLAGS + list(extra) + ["--cuda-device-only", "-S", srcp, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc -S failed for {src}:\n{r.stderr[-2000:]}")
    return open(out).read()


def kernels(asm):
    """name -> (instruction lines, metadata dict)."""
    out = {}
    cur, name = None, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if re.match(r"^\.Lfunc_end\d+:", line):       # (blocks may be laid out after the first s_endpgm)
                out[name] = [cur, {}]
                cur = None
                continue
            cur.append(line)
    for m in re.finditer(r"\.name:\s+(_Z\w+)\n((?:\s+\.\w+:.*\n)+)", asm):
        if m.group(1) in out:
            for kv in re.finditer(r"\.(\w+):\s+(\S+)", m.group(2)):
                out[m.group(1)][1][kv.group(1)] = kv.group(2)
    # metadata blocks list .name after other keys in some versions: fall back to a per-kernel scan of the amdhsa directives
    for name in out:
        m = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", asm, re.S)
        if m:
            for kv in re.finditer(r"\.amdhsa_(\w+)\s+(\S+)", m.group(1)):
                out[name][1]["amdhsa_" + kv.group(1)] = kv.group(2)
    return out


def vregs(text, acc=False):
    """set of VGPR indices an operand string mentions (acc: accumulation registers too, numbered from 1000)."""
    regs = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", text):
        kind = m.group(1) or m.group(4)
        if kind == "a" and not acc:
            continue
        off = 1000 if kind == "a" else 0
        if m.group(5) is not None:
            regs.add(off + int(m.group(5)))
        else:
            regs.update(range(off + int(m.group(2)), off + int(m.group(3)) + 1))
    return regs


def is_inst(line):
    s = line.strip()
    return bool(s) and not s.startswith((";", ".", "//")) and not s.endswith(":")


def scratch_bytes(meta):
    for k in ("amdhsa_private_segment_fixed_size", "private_segment_fixed_size"):
        if k in meta:
            return int(meta[k], 0)
    return None


def loop_headers(lines):
    return [m.group(1) for l in lines for m in [re.match(r"^\.LBB(\w+):.*Loop Header: Depth=1", l.strip())] if m]


def split_loop(lines, hdr=None):
    """(before, loop, after): the instructions of the kernel's first depth-1 loop, rotated to start at its header (the compiler
    may place the latch block in front of the header), and the straight-line code around it.  Loop membership comes from the
    block annotations (`in Loop: Header=BBx_y` / `Loop Header`) on labels and `; %bb.N:` comments."""
    if hdr is None:
        hs = loop_headers(lines)
        hdr = hs[0] if hs else None
    if hdr is None:
        return [(i, l.strip()) for i, l in enumerate(lines) if is_inst(l)], [], []
    before, loop_pre, loop_post, after = [], [], [], []
    inloop, seen_hdr = False, False
    for i, l in enumerate(lines):
        t = l.strip()
        if re.match(r"^\.LBB\w+:", t) or re.match(r"^; %bb\.\d+:", t):
            if re.match(r"^\.LBB" + hdr + r":", t):
                inloop, seen_hdr = True, True
            else:
                inloop = re.search(r"Header=BB" + hdr + r"\b", t) is not None
            continue
        if not is_inst(l):
            continue
        if inloop:
            (loop_post if seen_hdr else loop_pre).append((i, t))
        else:
            (after if seen_hdr else before).append((i, t))
    return before, loop_post + loop_pre, after


def check_inflight(name, lines):
    """No instruction touches the destination registers of an asm global_load while it may still be in flight."""
    errs = []
    before, loop, after = split_loop(lines)
    insts = before + loop + loop + after
    order = list(range(len(insts)))
    if not loop:
        errs.append(f"{name}: no loop found")
    queue = []          # in-order outstanding VM ops: set of destination VGPRs (empty for DMA / stores)
    pending_waits = []  # alternative waits of one branch diamond (only scalar code between them): the most lenient one applies
    nload = 0

    def flush():
        if pending_waits:
            n = max(pending_waits)
            del pending_waits[:]
            while len(queue) > n:
                queue.pop(0)

    for k in order:
        ln, text = insts[k]
        if text.endswith(":") or re.match(r"^\.LBB", text):
            continue
        m = re.match(r"^s_waitcnt\s+(.*)$", text)
        if m:
            v = re.search(r"vmcnt\((\d+)\)", m.group(1))
            if v:
                pending_waits.append(int(v.group(1)))
            continue
        op = text.split()[0]
        if op.startswith("s_") and not op.startswith("s_waitcnt"):
            continue                               # scalar code between alternative waits does not separate them
        flush()
        inflight = set().union(*queue) if queue else set()
        if re.match(r"^global_load_lds_", op) or re.match(r"^(global|buffer|scratch)_store", op) or op.startswith("global_atomic"):
            if vregs(text) & inflight:
                errs.append(f"{name}:{ln}: `{text}` uses a register with a load in flight")
            queue.append(set())
            continue
        if re.match(r"^(global|buffer|scratch)_load", op):
            ops = text[len(op):].split(",")
            dst, rest = vregs(ops[0]), vregs(",".join(ops[1:]))
            if (dst | rest) & inflight:
                errs.append(f"{name}:{ln}: `{text}` overlaps a register with a load in flight")
            queue.append(dst)
            nload += 1
            continue
        hit = vregs(text) & inflight
        if hit:
            errs.append(f"{name}:{ln}: `{text}` touches v{sorted(hit)} while its load may still be in flight")
    if nload == 0:
        errs.append(f"{name}: no global loads found (kernel not recognised)")
    return errs


def check_asm_reads_mfma(src, kdict):
    """An inline-asm VALU instruction must never be the FIRST reader of an MFMA result: the hazard recogniser cannot see inside
    an asm string, so no wait states are inserted behind the MFMA (round 3: a hand-written v_max3_f32 in the attention block's
    softmax read its scores before the matrix pipe had written them -- every tolerance test passed, the bitwise ones did not)."""
    errs = []
    for name, (lines, _) in kdict.items():
        fresh = set()          # registers written by an MFMA and not yet read by a compiler-scheduled instruction
        in_asm = False
        for ln, l in enumerate(lines):
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not is_inst(l):
                continue
            op = t.split()[0]
            ops = t[len(op):].split(",")
            if op.startswith("v_mfma"):
                fresh -= vregs(",".join(ops[1:]))       # (reads by another MFMA are interlocked by the hardware)
                fresh |= vregs(ops[0])
                continue
            reads = vregs(",".join(ops[1:])) if not op.startswith(("global_store", "ds_write", "buffer_store")) else vregs(t)
            if in_asm and op.startswith("v_") and reads & fresh:
                errs.append(f"{src}:{name}:{ln}: inline-asm `{t[:60]}` is the first reader of MFMA results v{sorted(reads & fresh)[:4]}")
            fresh -= vregs(t)
    return errs


def check_m0_nop(src, kdict):
    errs = []
    for name, (lines, _) in kdict.items():
        ins = [l.strip() for l in lines if is_inst(l)]
        for i, l in enumerate(ins):
            if l.startswith("global_load_lds_"):
                # (the builtin form is padded by the compiler's hazard recogniser: any predecessor that does not write M0 is fine)
                if ins[i - 1].startswith("s_mov_b32 m0"):
                    errs.append(f"{src}:{name}: LDS-DMA directly after `{ins[i - 1]}` (M0 hazard: one wait state needed)")
    return errs


def count_in_loop(lines, pat):
    """occurrences of an instruction pattern inside the kernel's outermost persistent loop (first loop header .. last back branch)."""
    best = 0
    for h in loop_headers(lines):
        _, loop, _ = split_loop(lines, h)
        best = max(best, sum(1 for _, l in loop if re.match(pat, l)))
    return best


def main():
    errs = []
    # ---- embed.hip, both element types (elem.h) ----
    for extra in ((), ("-DMIVIT_ELEM_F16",)):
        ke = kernels(device_asm("embed.hip", extra))
        d2 = [n for n in ke if "embed_fwd_direct2" in n]
        if not d2:
            errs.append(f"embed.hip {extra}: embed_fwd_direct2 not found")
        for n in d2:
            sb = scratch_bytes(ke[n][1])
            if sb is None or sb != 0:
                errs.append(f"{n} {extra}: scratch bytes = {sb} (must be 0: a spill of a load destination is invisible to the hand-placed waits)")
            errs += check_inflight(n, ke[n][0])
        errs += check_m0_nop("embed.hip", ke)
    # ---- the other users of the written-out DMA ----
    for src in ("fused_fwd.hip", "attention_fast.hip"):
        for extra in ((), ("-DMIVIT_ELEM_F16",), ("-DMIVIT_WIDTH64",), ("-DMIVIT_WIDTH64", "-DMIVIT_ELEM_F16")):
            if src == "attention_fast.hip" and "-DMIVIT_WIDTH64" in extra:
                continue
            errs += check_asm_reads_mfma(src, kernels(device_asm(src, extra)))
    for src, extra in (("rowstream.hip", ()), ("gemm_dma.hip", ()), ("wgrad_dma.hip", ()), ("fused_bwd.hip", ()),
                       ("fused_bwd.hip", ("-DMIVIT_WIDTH64",))):
        kd = kernels(device_asm(src, extra))
        errs += check_m0_nop(src, kd)
        errs += check_asm_reads_mfma(src, kd)
        for n, (lines, meta) in kd.items():
            sb = scratch_bytes(meta)
            # (scratch traffic in these kernels is a performance matter only: extra VM operations can only make a counted wait
            #  stricter, never looser)
            if "mlp_block_bwd" in n or "attn_out_bwd" in n:
                c = count_in_loop(lines, r"^global_store_dwordx2\b")
                # (the eight-wave kernel, and both kernels at width 64: one row store per row tile and wave -- NROWST in the source)
                need = 2 if "mlp_block_bwd8" in n or extra else 4
                if c < need:
                    errs.append(f"{n}: {c} global_store_dwordx2 in the tile loop; wait_vm<{need}> counts on >= {need} row stores per tile")
            if "qkv_bwd_kernel" in n:
                c = count_in_loop(lines, r"^global_store_dwordx2\b")
                if c < 2:
                    errs.append(f"{n}: {c} global_store_dwordx2 in the tile loop; wait_vm<2> counts on 2 row stores per tile")
            if "rowstream_kernel" in n:
                c = count_in_loop(lines, r"^global_store_dwordx4\b")
                m = re.search(r"rowstream_kernelILi(\d+)", n)
                need = 1 if m and int(m.group(1)) > 256 else 2
                if c < need:
                    errs.append(f"{n}: {c} 16-byte stores in the tile loop; the tile wait counts on >= {need} per epilogue")
    if errs:
        print("\n".join(errs))
        return 1
    print("[isa_check] all invariants hold")
    return 0


if __name__ == "__main__":
    sys.exit(main())
