#!/usr/bin/env python3
"""LDS (default) vs LDS-free (sgpr) MeanShift iteration kernel on the headline vote batch (64 frames x 9 fits), alone."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import make_inputs, run_postproc, StageTimer
from pvn3d_amd.lib.utils import _vote_engine as eng
dev = torch.device("cuda:0")
inp = make_inputs(64, 12288, 3072, dev, 0)
off = StageTimer(False)
for kern in (None, "sgpr", "sgpr+cap4096", "sgpr+cap2048", "sgpr+cap1024", "sgpr+cap512"):
    eng.DEFAULT_KERNEL = kern
    for _ in range(3):
        run_postproc(inp, off, 4)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); run_postproc(inp, off, 4); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    print("%-14s vote->pose %.2f ms" % (kern, float(np.median(ts))), flush=True)
