#!/usr/bin/env python3
"""B = 1 Pointnet2MSG eval forward, a few calls (for rocprofv3 --kernel-trace): python tools/b1_trace.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd import synth  # noqa: E402
from pvn3d_amd.lib.pointnet2_msg import Pointnet2MSG  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = Pointnet2MSG(input_channels=6).to(dev).eval()
f = synth.synth_frame(frame=7000, n_pts=12288, n_obj=3072)
pc = torch.from_numpy(np.concatenate([f["pcld"], f["feats"].T], 1)[None]).to(dev)
with torch.no_grad():
    for _ in range(6):
        net(pc)
torch.cuda.synchronize()
