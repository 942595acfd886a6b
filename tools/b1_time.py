#!/usr/bin/env python3
"""B = 1 Pointnet2MSG eval forward: eager and HIP-graph replay latency (median of 30)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd import synth
from pvn3d_amd.lib.pointnet2_msg import Pointnet2MSG
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = Pointnet2MSG(input_channels=6).to(dev).eval()
f = synth.synth_frame(frame=7000, n_pts=12288, n_obj=3072)
pc = torch.from_numpy(np.concatenate([f["pcld"], f["feats"].T], 1)[None]).to(dev)
def med(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
with torch.no_grad():
    e = med(lambda: net(pc))
    g = net.graphed(pc)
    r = med(lambda: g(pc))
print("B=1 forward: eager %.3f ms, graph replay %.3f ms" % (e, r))
