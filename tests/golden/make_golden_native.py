#!/usr/bin/env python3
"""Generate tests/golden/native_ref.npz with the REFERENCE'S OWN native-op kernels.

The kernels of pvn3d/_ext-src/src/*_gpu.cu are compiled for the CPU by oracle/ref_shim/build_ref.py
(-> oracle/_ref/, see that file for how) and run here on the BASELINE shapes: the four
set-abstraction levels and the four feature-propagation levels of the PVN3D backbone
(lib/pvn3d.py:67-118) on two 12 288-point clouds (a plain one and one with 10 % 'wrap'-padded
duplicate points, linemod_dataset.py:264, where exact distance ties are common).

Stored per cloud c in {0,1} and level l in {0..3}:
  c{c}_xyz                      the input cloud (float32)
  c{c}_fps{l}                   furthest_point_sampling(xyz_l, npoint_l)            int16
  c{c}_bq{l}_{s}                ball_query(new_xyz_l, xyz_l, radius_{l,s}, nsample) int16
  c{c}_nn{l}_idx / _d2          three_nn(unknown = xyz_l, known = xyz_{l+1})        int16 / float32
and, per op, how many of those outputs change when the same sources are built with
floating-point contraction (-ffp-contract=fast -mfma): `fma_flips` (the reference's CUDA binary was
built with nvcc's default -fmad=true, which may contract; it cannot be observed here).

Needs /root/reference + g++ (the build container).  The committed fixture is what tests read.
Usage: python tests/golden/make_golden_native.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from oracle import ref  # noqa: E402
from pvn3d_amd import synth  # noqa: E402

NPOINT = [2048, 1024, 512, 128]                                   # lib/pvn3d.py:67-111
RADII = [(0.0175, 0.025), (0.025, 0.05), (0.05, 0.1), (0.1, 0.2)]
NSAMPLE = (16, 32)


def main():
    assert ref.build(), "needs /root/reference"
    out = {}
    flips = {"fps": [0, 0], "ball_query": [0, 0], "three_nn_idx": [0, 0], "three_nn_d2": [0, 0]}

    def tally(key, a, b):
        flips[key][0] += int((a != b).sum())
        flips[key][1] += int(a.size)

    for c, kw in enumerate((dict(frame=0), dict(frame=1, wrap_pad=0.10))):
        xyz = synth.synth_frame(n_pts=12288, n_obj=3072, **kw)["pcld"][None]
        out["c%d_xyz" % c] = xyz[0]
        levels = [xyz]
        for l in range(4):
            cur = levels[-1]
            fps = ref.furthest_point_sampling(cur, NPOINT[l])
            tally("fps", fps, ref.furthest_point_sampling(cur, NPOINT[l], variant="fma"))
            out["c%d_fps%d" % (c, l)] = fps[0].astype(np.int16)
            new = np.ascontiguousarray(cur[:, fps[0]])
            for s in range(2):
                bq = ref.ball_query(new, cur, RADII[l][s], NSAMPLE[s])
                tally("ball_query", bq, ref.ball_query(new, cur, RADII[l][s], NSAMPLE[s], variant="fma"))
                out["c%d_bq%d_%d" % (c, l, s)] = bq[0].astype(np.int16)
            levels.append(new)
        for l in range(4):
            d2, idx = ref.three_nn(levels[l], levels[l + 1])
            d2f, idxf = ref.three_nn(levels[l], levels[l + 1], variant="fma")
            tally("three_nn_idx", idx, idxf)
            tally("three_nn_d2", d2, d2f)
            out["c%d_nn%d_idx" % (c, l)] = idx[0].astype(np.int16)
            out["c%d_nn%d_d2" % (c, l)] = d2[0]
        print("cloud %d done" % c, flush=True)
    out["fma_flips"] = np.frombuffer(json.dumps(flips).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "native_ref.npz"), **out)
    print("fma sensitivity (changed, total):", flips)
    print("wrote native_ref.npz, %d KiB" % (os.path.getsize(os.path.join(HERE, "native_ref.npz")) // 1024))


if __name__ == "__main__":
    main()
