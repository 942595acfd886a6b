#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of the HIP translation units (no GPU needed).

Compiles each pvn3d_amd/csrc/*.hip device-only with -Rpass-analysis=kernel-resource-usage using the
flags of pvn3d_amd/csrc/Makefile and prints one row per kernel.  `--json FILE` stores the table
(profiles/rNN_resource_usage.json); `--fail-on-scratch PATTERN` exits 1 if a kernel whose demangled
name matches PATTERN uses scratch memory (register spills).
Usage: python tools/resource_usage.py [file.hip ...] [--json out.json] [--fail-on-scratch mlp_chain]
"""
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pvn3d_amd", "csrc")
KEYS = {"VGPRs": "vgpr", "AGPRs": "agpr", "SGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch",
        "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds_static",
        "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill"}


def makefile_flags(stem):
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^HIPFLAGS\s*=\s*(.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    extra = re.search(r"^EXTRA_%s\s*=\s*(.*)$" % re.escape(stem), mk, re.M)
    return flags + (extra.group(1).split() if extra else [])


def analyse(path):
    stem = os.path.splitext(os.path.basename(path))[0]
    cmd = ["/opt/rocm/bin/hipcc"] + makefile_flags(stem) + ["--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                                                            "-c", path, "-o", "/dev/null"]
    err = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True, check=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = {"file": os.path.basename(path), "mangled": body.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            if k.strip() in KEYS:
                cur[KEYS[k.strip()]] = int(v)
    if rows:
        dem = subprocess.run(["c++filt"] + [r["mangled"] for r in rows],
                             stdout=subprocess.PIPE, text=True).stdout.splitlines()
        for r, d in zip(rows, dem):
            d = re.sub(r"\(anonymous namespace\)::", "", d)
            r["kernel"] = re.sub(r"\(.*$", "", re.sub(r"^void ", "", d))
    return rows


def main(argv):
    files, out_json, pat = [], None, None
    it = iter(argv)
    for a in it:
        if a == "--json":
            out_json = next(it)
        elif a == "--fail-on-scratch":
            pat = next(it)
        else:
            files.append(a)
    files = files or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    rows = [r for f in files for r in analyse(os.path.abspath(f))]
    print("%-16s %-58s %5s %5s %8s %4s" % ("file", "kernel", "VGPR", "AGPR", "scratch", "occ"))
    for r in rows:
        print("%-16s %-58s %5d %5d %8d %4d" % (r["file"], r["kernel"][:58], r.get("vgpr", -1), r.get("agpr", -1),
                                               r.get("scratch", -1), r.get("occupancy", -1)))
    if out_json:
        with open(out_json, "w") as fh:
            json.dump({"flags": "pvn3d_amd/csrc/Makefile HIPFLAGS, gfx950", "kernels": rows}, fh, indent=1)
    bad = [r["kernel"] for r in rows if pat and re.search(pat, r["kernel"]) and r.get("scratch", 0) > 0]
    if bad:
        print("kernels with scratch:", bad)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
