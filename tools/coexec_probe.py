#!/usr/bin/env python3
"""Co-execution probe: the 64-frame Pointnet2MSG forward (fp32-MFMA kernels) beside a VALU-only kernel on a second
stream.  If the pair takes max(A, B) the matrix and vector pipes co-execute; if it takes A + B they do not."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import make_inputs  # noqa: E402
from pvn3d_amd.lib.pointnet2_msg import Pointnet2MSG  # noqa: E402

dev = torch.device("cuda:0")
probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcoexec_probe.so"))
probe.coexec_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
torch.manual_seed(0)
net = Pointnet2MSG(input_channels=6).to(dev).eval()
inp = make_inputs(64, 12288, 3072, dev, 0)
pc = torch.cat([inp["pcld"], inp["feats"].transpose(1, 2)], 2).contiguous()
out = torch.zeros(256, device=dev)
s2 = torch.cuda.Stream()


def med(fn, n=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def run_a():
    with torch.no_grad():
        net(pc)


for mode in (0, 1):
    for blocks in (256, 512, 1024):
        iters = 60000 * 2048 // blocks // 4
        def run_b():
            probe.coexec_launch(mode, blocks, iters, out.data_ptr(), s2.cuda_stream)
        def both():
            run_b(); run_a()
        a, b, ab = med(run_a), med(run_b), med(both)
        print("mode %d (%s) blocks %5d: MLP forward %.2f ms, VALU kernel %.2f ms, both %.2f ms  (sum %.2f, max %.2f)"
              % (mode, "pure VALU" if mode == 0 else "VALU + LDS broadcast reads", blocks, a, b, ab, a + b, max(a, b)), flush=True)
