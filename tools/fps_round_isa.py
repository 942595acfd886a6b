#!/usr/bin/env python3
"""Instruction count of the FPS round body (csrc/fps_cells.hip, fps_cells_kernel<3>: 4096 < n <= 12288) from the
compiler's ISA: the kernel's basic blocks with their loop depth and instruction mix, i.e. what a sampling round can
execute.  No GPU needed.  Usage: python tools/fps_round_isa.py [out.txt]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def kind(i):
    if i.startswith("v_"):
        return "valu"
    if i.startswith("s_cbranch") or i == "s_branch":
        return "branch"
    if i.startswith("s_waitcnt") or i == "s_nop":
        return "wait"
    if i.startswith("s_"):
        return "salu"
    if i.startswith("ds_"):
        return "lds"
    return "vmem"


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    src = os.path.join(ROOT, "pvn3d_amd", "csrc", "fps_cells.hip")
    asm = "/tmp/fps_cells_isa.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-Wno-inline-asm", "-S", "--cuda-device-only", src, "-o", asm], cwd=os.path.dirname(src),
                          stderr=subprocess.DEVNULL)
    L = open(asm).read().splitlines()
    s = next(i for i, l in enumerate(L) if re.match(r"^_ZN.*fps_cells_kernelILi3E.*:", l))
    e = next(i for i in range(s, len(L)) if "s_endpgm" in L[i])
    blocks, name, depth, cur = [], "entry", 0, []
    for l in L[s + 1:e + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((name, depth, cur))
            name, cur = m.group(1), []
            d = re.search(r"Depth=(\d+)", l)
            depth = int(d.group(1)) if d else 0
            continue
        d = re.search(r";\s+in Loop: Header=\S+ Depth=(\d+)", l)
        if d and not cur:
            depth = int(d.group(1))
        t = l.strip()
        if t and not t.startswith(";") and not t.startswith("."):
            cur.append(t.split()[0])
    blocks.append((name, depth, cur))
    lines = ["fps_cells_kernel<3> (n in 4097..12288): %d instructions in %d basic blocks"
             % (sum(len(b[2]) for b in blocks), len(blocks))]
    in_loop = [b for b in blocks if b[1] >= 1 and b[2]]
    kinds = collections.Counter()
    for _, _, ins in in_loop:
        kinds.update(kind(i) for i in ins)
    lines.append("blocks inside the sample loop (loop depth >= 1): %d blocks, %d instructions in all -- every round executes a "
                 "subset (the common path skips the cell-refresh, tie and exhausted-cell blocks)"
                 % (len(in_loop), sum(len(b[2]) for b in in_loop)))
    lines.append("mix of those: " + ", ".join("%s %d" % kv for kv in kinds.most_common()))
    lines.append("")
    lines.append("%-12s %5s %6s  %s" % ("block", "depth", "instr", "valu / salu / branch / wait / lds / vmem"))
    for name, depth, ins in in_loop:
        c = collections.Counter(kind(i) for i in ins)
        lines.append("%-12s %5d %6d  %d / %d / %d / %d / %d / %d" % (name, depth, len(ins), c["valu"], c["salu"], c["branch"],
                                                                     c["wait"], c["lds"], c["vmem"]))
    text = "\n".join(lines)
    print(text)
    if out:
        with open(out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
