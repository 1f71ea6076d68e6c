#!/usr/bin/env python3
"""Static instruction mix of device kernels, per kernel and per basic block, from the compiler's gfx950 assembly.

    scripts/isa_mix.py [kernel-name-substring ...]      (default: k_smooth_tile k_geom_tile)

Compiles csrc/smgpu.hip with the build's own flags to assembly (device side only, into /tmp) and classifies every
instruction of the kernels whose demangled name contains one of the substrings:

    f64      v_*_f64 arithmetic (add / mul / fma / div_fixup / div_fmas / div_scale / rcp / rsq / sqrt / min / max / ldexp / frexp / fract ...)
    cmp      v_cmp* (any type)
    sel      v_cndmask / v_readlane / v_readfirstlane / v_writelane / v_permlane / ds_bpermute-free selects
    mov      v_mov / v_accvgpr
    int      every other v_* (address arithmetic, table decode, masks, conversions)
    lds      ds_*
    vmem     global_* / buffer_* / flat_* / scratch_*
    salu     s_* except s_waitcnt / s_nop / s_barrier / branches
    ctl      s_waitcnt, s_nop, s_barrier, branches, s_endpgm

Per basic block: the label, the counts and -- where the block jumps back to itself or to an earlier label -- a "loop" mark, so that
the blocks with the reference's arithmetic, the ELL decode and the LDS addressing can be told apart.  The dynamic counterpart
(what a launch executes, by type) comes from the SQ_INSTS_VALU_* counters: scripts/pmc_inst_mix.sh.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "smoothmesh_amd", "csrc", "smgpu.hip")
ASM = "/tmp/smgpu_isa_mix.s"

CLASSES = ["f64", "cmp", "sel", "mov", "int", "lds", "vmem", "salu", "ctl"]


def classify(op):
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_") and "_f64" in op and not op.startswith("v_cvt"):
        return "f64"
    if op.startswith(("v_cndmask", "v_readlane", "v_readfirstlane", "v_writelane", "v_permlane")):
        return "sel"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")):
        return "mov"
    if op.startswith("v_"):
        return "int"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_branch", "s_cbranch", "s_endpgm", "s_setprio", "s_sleep", "s_setpc", "s_swappc")):
        return "ctl"
    if op.startswith("s_"):
        return "salu"
    return None


def flags():
    mk = open(os.path.join(ROOT, "smoothmesh_amd", "csrc", "Makefile")).read()
    m = re.search(r"^HIPFLAGS\s*[:?]?=\s*(.*)$", mk, re.M)
    return m.group(1).split() if m else ["-O3", "-std=c++17", "-ffp-contract=off"]


def main():
    pats = sys.argv[1:] or ["k_smooth_tile", "k_geom_tile"]
    fl = [f for f in flags() if not f.startswith(("-shared", "-o", "-fPIC"))]
    if "--offload-arch=gfx950" not in fl:
        fl.append("--offload-arch=gfx950")
    cmd = ["/opt/rocm/bin/hipcc"] + fl + ["--cuda-device-only", "-S", SRC, "-o", ASM]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr[-2000:])
    cur = None
    kernels = collections.OrderedDict()
    for line in open(ASM):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        kernels[cur].append(line.rstrip("\n"))
    names = subprocess.run(["c++filt"], input="\n".join(kernels), capture_output=True, text=True).stdout.split("\n")
    for mangled, name in zip(kernels, names):
        short = name.split("(")[0].replace("void ", "").replace("smgpu::", "")
        if not any(p in short for p in pats) or not kernels[mangled]:
            continue
        blocks = collections.OrderedDict()
        order = {}
        label = "entry"
        blocks[label] = collections.Counter()
        order[label] = 0
        back = set()
        for line in kernels[mangled]:
            m = re.match(r"^(\.LBB\w+):", line)
            if m:
                label = m.group(1)
                blocks[label] = collections.Counter()
                order[label] = len(order)
                continue
            t = line.strip()
            if not t or t.startswith((";", ".", "//")):
                continue
            op = t.split()[0]
            c = classify(op)
            if c is None:
                continue
            blocks[label][c] += 1
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = t.split()[-1]
                if tgt in order and order[tgt] <= order[label]:
                    back.add(tgt)
                    blocks[label]["_back"] = 1
        total = collections.Counter()
        for b in blocks.values():
            total.update({k: v for k, v in b.items() if not k.startswith("_")})
        n = sum(total.values())
        print(f"== {short}   {n} instructions, {len(blocks)} blocks")
        print("   " + "  ".join(f"{c} {total[c]} ({100.0 * total[c] / max(n, 1):.0f}%)" for c in CLASSES))
        print("   blocks with >= 40 instructions (L = the target of a backward branch: a loop head; B = ends in a backward branch):")
        for lab, b in blocks.items():
            nb = sum(v for k, v in b.items() if not k.startswith("_"))
            if nb < 40:
                continue
            mark = ("L" if lab in back else " ") + ("B" if b.get("_back") else " ")
            print(f"     {lab:14s} {mark} {nb:5d}  " + "  ".join(f"{c} {b[c]}" for c in CLASSES if b[c]))


if __name__ == "__main__":
    main()
