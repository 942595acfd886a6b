#!/usr/bin/env python3
"""Recipe for oracle/_ref: the reference's OWN native-op kernels, compiled for the CPU.

TEST INFRASTRUCTURE (used by tests/ and tests/golden/make_golden_native.py only).

The nine native ops of ethnhe/PVN3D exist only as CUDA (`pvn3d/_ext-src/src/*_gpu.cu`); there is no
nvcc and no NVIDIA GPU here.  Their kernels are plain C++ apart from the CUDA execution model, so
this recipe compiles the reference's source files *where they lie* under /root/reference with g++:

  * `oracle/ref_shim/include/` stands in for <cuda.h>, <cuda_runtime.h>, <ATen/...> and supplies
    threadIdx/blockIdx/blockDim/gridDim, __global__/__device__/__shared__, min/max, atomicAdd,
    __syncthreads and at::cuda::getCurrentCUDAStream (cuda_cpu_shim.h).  The reference's own
    `include/cuda_utils.h` (opt_n_threads, opt_block_config, CUDA_CHECK_ERRORS) is used unchanged.
  * the only thing g++ cannot parse is the `kernel<<<grid, block, shmem, stream>>>(args)` launch
    syntax.  This script rewrites exactly that token pattern into `shim::launch(shim::LaunchCfg(grid,
    block, shmem, stream), kernel, args)` while copying each .cu to a build file under
    oracle/_ref/gen/ (git-ignored, never committed); no other byte of the source is touched, which
    the script asserts.
  * threads of a block run as cooperative fibers, so the FPS kernel's shared-memory halving tree and
    its nine __syncthreads per round execute literally (cuda_cpu_shim.cpp).

Two libraries are produced, differing only in floating-point contraction:
  oracle/_ref/libpvn3d_ref_nofma.so   g++ -ffp-contract=off          (C-source semantics, one rounding per op)
  oracle/_ref/libpvn3d_ref_fma.so     g++ -ffp-contract=fast -mfma   (what a contracting compiler such as
                                                                      nvcc's default -fmad=true may emit)
because the reference's CUDA 9 binary cannot be observed here (tests report how the two differ).

Outputs go only to oracle/_ref/.  Nothing is copied into the repository.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "_ref")
REF = os.environ.get("PVN3D_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "pvn3d", "_ext-src", "src")
INC = os.path.join(REF, "pvn3d", "_ext-src", "include")
CU = ["sampling_gpu.cu", "ball_query_gpu.cu", "group_points_gpu.cu", "interpolate_gpu.cu"]

LAUNCH = re.compile(r"([A-Za-z_]\w*(?:\s*<\s*\w+\s*>)?)\s*<<<(.*?)>>>\s*\(", re.S)


def rewrite_launches(text):
    """kernel<<<cfg>>>(args) -> shim::launch(shim::LaunchCfg(cfg), kernel, args)."""
    out, n = LAUNCH.subn(lambda m: "shim::launch(shim::LaunchCfg(%s), %s, " % (m.group(2), m.group(1)), text)
    # everything outside the launch tokens is byte-identical
    assert LAUNCH.sub("", text).replace(" ", "") == re.sub(
        r"shim::launch\(shim::LaunchCfg\(.*?\), [A-Za-z_]\w*(?:\s*<\s*\w+\s*>)?, ", "", out, flags=re.S
    ).replace(" ", "")
    assert "<<<" not in out and n > 0
    return out, n


def have_reference():
    return all(os.path.isfile(os.path.join(SRC, f)) for f in CU) and os.path.isfile(os.path.join(INC, "cuda_utils.h"))


def build(force=False):
    if not have_reference():
        print("build_ref: %s not present; keeping whatever oracle/_ref already holds" % REF)
        return False
    gen = os.path.join(OUT, "gen")
    os.makedirs(gen, exist_ok=True)
    gens = []
    for f in CU:
        with open(os.path.join(SRC, f)) as fh:
            text, n = rewrite_launches(fh.read())
        dst = os.path.join(gen, f.replace(".cu", ".cpp"))
        if not os.path.exists(dst) or open(dst).read() != text:
            with open(dst, "w") as fh:
                fh.write(text)
        gens.append(dst)
    common = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-w",
              "-I", os.path.join(HERE, "include"), "-I", INC]
    srcs = gens + [os.path.join(HERE, "cuda_cpu_shim.cpp"), os.path.join(HERE, "ref_exports.cpp"),
                   os.path.join(HERE, "ref_opt.cpp")]
    for name, flags in (("nofma", ["-ffp-contract=off"]), ("fma", ["-ffp-contract=fast", "-mfma"])):
        so = os.path.join(OUT, "libpvn3d_ref_%s.so" % name)
        newest = max(os.path.getmtime(s) for s in srcs + [__file__])
        if force or not os.path.exists(so) or os.path.getmtime(so) < newest:
            subprocess.check_call(common + flags + srcs + ["-o", so])
    return True


if __name__ == "__main__":
    ok = build(force="-B" in sys.argv)
    sys.exit(0 if ok or os.path.exists(os.path.join(OUT, "libpvn3d_ref_nofma.so")) else 1)
