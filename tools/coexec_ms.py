#!/usr/bin/env python3
"""Island A (64-frame Pointnet2MSG forward) beside island B (vote -> MeanShift -> pose of 64 frames) on two streams:
A alone, B alone, both (either enqueue order), for the LDS and the LDS-free MeanShift kernels."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import make_inputs, run_postproc, run_net, make_net, StageTimer
from pvn3d_amd.lib.utils import _vote_engine as eng
dev = torch.device("cuda:0")
inp = make_inputs(64, 12288, 3072, dev, 0)
inp["pc"] = torch.cat([inp["pcld"], inp["feats"].transpose(1, 2)], 2).contiguous()
net = make_net(dev)
off = StageTimer(False)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
poll = int(sys.argv[1]) if len(sys.argv) > 1 else 4

def med(fn, n=7):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))

def a():
    with torch.cuda.stream(s1):
        return run_net(net, inp, off)
def b():
    with torch.cuda.stream(s2):
        return run_postproc(inp, off, poll)
ta = med(a)
for kern in (None, "sgpr", "sgpr+cap2048", "sgpr+cap1024"):
    eng.DEFAULT_KERNEL = kern
    tb = med(b)
    tab = med(lambda: (a(), b()))
    tba = med(lambda: (b(), a()))
    print("%-13s A %.2f  B %.2f  A-then-B %.2f  B-then-A %.2f  (sum %.2f)" % (kern, ta, tb, tab, tba, ta + tb), flush=True)
