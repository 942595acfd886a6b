#!/usr/bin/env python3
"""How many MeanShift seeds are bitwise fixed points per iteration (heavy-tailed vote set of bench.py)?
Build: (cd pvn3d_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMS_FROZEN_PROBE -c meanshift.hip -o /tmp/ms_probe.o
        && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libpvn3d_msprobe.so $(ls *.o | grep -v '^meanshift.o$') /tmp/ms_probe.o)
Run:   PVN3D_HIP_LIB=tools/libpvn3d_msprobe.so python tools/ms_frozen_stats.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd import synth  # noqa: E402
from pvn3d_amd._lib import lib  # noqa: E402
from pvn3d_amd.lib.utils import pvn3d_eval_utils as ev  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    fr = [synth.synth_frame(frame=7400 + i, n_pts=12288, n_obj=3072, outlier_frac=0.10, sig_out=0.30) for i in range(4)]
    st = lambda k: torch.from_numpy(np.stack([f[k] for f in fr], 0)).to(dev)
    buf = (ctypes.c_int * 1024)()
    lib.pvn3d_ms_probe_read(buf, 1)
    res = ev.cal_batch_poses_lm(st("pcld").contiguous(), st("mask").to(torch.int32).contiguous(), st("ctr_of").contiguous(),
                                st("pred_kp_of").contiguous(), True, 2, False, 1, poll_every=4)
    torch.cuda.synchronize()
    print("iters", res["iters"].cpu().numpy().tolist())
    lib.pvn3d_ms_probe_read(buf, 0)
    a = np.array(buf[:])
    for t in list(range(1, 40)) + list(range(40, 300, 10)):
        if a[512 + t]:
            print("t=%3d  iterated %6d  fixed points %6d  (%.1f %%)" % (t, a[512 + t], a[t], 100.0 * a[t] / a[512 + t]))


if __name__ == "__main__":
    main()
