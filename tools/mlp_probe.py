#!/usr/bin/env python3
"""Phase cycle stamps of the fused MLP kernels (needs a -DSM_PROBE build, PVN3D_HIP_LIB=...).
Runs each SA/FP chain of Pointnet2MSG alone and prints, averaged over ALL workgroups of the launch,
the cycles spent up to each stamp since the previous one: 14 body start, 0 run start, 7 first layer-0 chunk staged, 8 layer-0 steady loop done, 6 (column-sliced kernel) first input chunk in LDS, 1/3/5 after layer l's MMA (+barrier),
2/4 after the activation store, 15 end."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["PVN3D_GEOMETRY_STREAM"] = "0"
from bench import make_inputs  # noqa: E402
from pvn3d_amd._lib import lib  # noqa: E402
from pvn3d_amd.lib.pointnet2_msg import Pointnet2MSG  # noqa: E402
from pvn3d_amd.lib.pointnet2_utils import _ext  # noqa: E402

lib.pvn3d_debug_mlp_probe_read.restype = ctypes.c_int
lib.pvn3d_debug_mlp_probe_read.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
inp = make_inputs(64, 12288, 3072, dev, 0)
net = Pointnet2MSG(input_channels=6).to(dev).eval()
pc = torch.cat([inp["pcld"], inp["feats"].transpose(1, 2)], 2).contiguous()
names = []
orig_sa, orig_fp = _ext.sa_mlp_maxpool, _ext.fp_interp_mlp


def report(tag):
    torch.cuda.synchronize()
    buf = np.zeros(32, dtype=np.uint64)
    lib.pvn3d_debug_mlp_probe_read(buf.ctypes.data)
    b = buf.astype(np.float64)
    n_wg = max(b[16 + 15], 1.0)
    order = [14, 0, 6, 7, 8, 1, 2, 3, 4, 5, 15]
    segs = ["->%d:%7.0f" % (k, b[k] / n_wg) for k in order if b[16 + k] > 0]
    tot = sum(b[k] for k in order if k != 14) / n_wg
    print("%-30s wgs %6d  cycles/wg %8.0f  %s" % (tag, int(n_wg), tot, "  ".join(segs)))


def sa(*a, **k):
    r = orig_sa(*a, **k)
    report("SA dims=%s ns=%d" % (a[5].dims, a[3].size(2)))
    return r


def fp(*a, **k):
    r = orig_fp(*a, **k)
    report("FP dims=%s" % (a[4].dims,))
    return r


with torch.no_grad():
    net(pc)
    torch.cuda.synchronize()
    lib.pvn3d_debug_mlp_probe_read(np.zeros(32, dtype=np.uint64).ctypes.data)     # reset the accumulators
    _ext.sa_mlp_maxpool, _ext.fp_interp_mlp = sa, fp
    net(pc)
