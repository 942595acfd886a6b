#!/usr/bin/env python3
"""Lint the gfx950 ISA of every translation unit for the packed-fp32 operand form that goes wrong on MI355X.

Measured (round 5; tools/sg_fault_repro.hip variant 20, tools/ms_beside_mfma.py; profiles/r05_pk_opsel_fault.txt): a
VOP3P packed-fp32 instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) whose LOW result half takes the HIGH register
of a VGPR pair in its src1 or src2 position -- op_sel:[_,1] / op_sel:[_,_,1] -- reads that operand as +0 in lanes 48-63
now and then while the other wave of its SIMD is inside an MFMA / LDS K loop.  Never observed for: the src0 position,
the opposite selection (op_sel_hi = 0), SGPR pairs.  hipcc emits the bad form by itself whenever a scalar that lives in
an odd register is broadcast to both halves (e.g. `float2{w, w} * pair` with w = some_float4.y).

No GPU needed: compiles each pvn3d_amd/csrc/*.hip device-only to assembly with the Makefile's flags and lists every
offending instruction (kernel, line).  Exit status 1 if there is one.  tests/test_abi.py runs it.
Usage: python tools/pk_opsel_lint.py [file.hip ...]
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pvn3d_amd", "csrc")
PK = re.compile(r"^\s*(v_pk_(?:mul|add|fma|max|min)_f32)\s+(.*)$")
SEL = re.compile(r"op_sel:\[([01,]+)\]")


def makefile_flags(stem):
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^HIPFLAGS\s*=\s*(.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    extra = re.search(r"^EXTRA_%s\s*=\s*(.*)$" % re.escape(stem), mk, re.M)
    return flags + (extra.group(1).split() if extra else [])


def offending(asm_text):
    """-> [(kernel, line_no, instruction)] for packed-fp32 instructions with op_sel bit 1 or 2 set on a VGPR source."""
    out, kernel = [], "?"
    for no, line in enumerate(asm_text.splitlines(), 1):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            kernel = m.group(1)
            continue
        m = PK.match(line)
        if not m:
            continue
        ops = m.group(2)
        sel = SEL.search(ops)
        if not sel:
            continue
        bits = [int(b) for b in sel.group(1).split(",")]
        operands = [o.strip() for o in ops.split(" op_sel")[0].split(",")]
        # operands[0] = vdst, [1] = src0, [2] = src1, [3] = src2 (v_pk_fma_f32); vector tuples are written v[a:b]
        # (their comma-free form survives the split)
        for pos in (1, 2):
            if pos < len(bits) and bits[pos] == 1 and pos + 1 < len(operands) and operands[pos + 1].startswith("v"):
                out.append((kernel, no, line.strip()))
                break
    return out


def lint_file(path):
    stem = os.path.splitext(os.path.basename(path))[0]
    with tempfile.TemporaryDirectory() as td:
        s_path = os.path.join(td, stem + ".s")
        cmd = ["/opt/rocm/bin/hipcc"] + makefile_flags(stem) + ["--cuda-device-only", "-S", path, "-o", s_path]
        subprocess.run(cmd, cwd=CSRC, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return offending(open(s_path).read())


def main(argv):
    from concurrent.futures import ThreadPoolExecutor
    files = [os.path.abspath(f) for f in argv] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    bad = 0
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        results = list(ex.map(lint_file, files))
    for f, hits in zip(files, results):
        print("%-22s %s" % (os.path.basename(f), "ok" if not hits else "%d packed-fp32 instruction(s) with a high->low "
                            "select on a VGPR src1/src2" % len(hits)))
        for kernel, no, ins in hits[:8]:
            print("    %s  line %d: %s" % (kernel[:60], no, ins))
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
