#!/usr/bin/env python3
"""Cycles per phase of a sampling round of the multi-wave FPS kernel (csrc/fps_cells.hip, -DPVN3D_FC_PROF build):
PVN3D_HIP_LIB=tools/libpvn3d_probe.so python tools/fps_prof.py [n] [m]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd import synth  # noqa: E402
from pvn3d_amd._lib import lib, check  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    dev = torch.device("cuda:0")
    xyz = torch.from_numpy(synth.synth_cloud(np.random.default_rng(1234), n)[0][None]).to(dev)
    words = lib.pvn3d_fps_ws_words(n)
    ws = torch.zeros((1, words), dtype=torch.int32, device=dev)
    out = torch.zeros((1, m), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        check(lib.pvn3d_furthest_point_sampling_ws_waves(1, n, m, xyz.data_ptr(), ws.data_ptr(), out.data_ptr(), None, None, 0,
                                                         3, st), "fps")
    torch.cuda.synchronize()
    acc = ws[0, :32].cpu().numpy().astype(np.int64).reshape(4, 8)
    names = ["cull", "own+refresh partials, publish", "update-only cells", "wait", "combine, cache write", "arg-max, results",
             "polls that failed", "-"]
    for w in range(3):
        tot = acc[w, :6].sum()
        print("wave %d: %.0f cycles per round = " % (w, tot / (m - 1)) +
              "  ".join("%s %.0f" % (names[p], acc[w, p] / (m - 1)) for p in range(7)))


if __name__ == "__main__":
    main()
