#!/usr/bin/env python
"""Replays the single-frame YCB vote -> cluster -> pose call (HIP graph and polled) a few times so that
`rocprofv3 --kernel-trace` shows the launches of one call and the gaps between them.  Prints the wall time per call."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvn3d_amd import synth
from pvn3d_amd.lib.utils import pvn3d_eval_utils as ev

dev = torch.device("cuda:0")
fy = [synth.synth_frame_ycb(frame=7200)]
sty = lambda k: torch.from_numpy(np.stack([f[k] for f in fy], 0)).to(dev)
yp, ym, yc, yk = sty("pcld").contiguous(), sty("mask").to(torch.int32).contiguous(), sty("ctr_of").contiguous(), sty("pred_kp_of").contiguous()
gy = ev.GraphedFramePoses("ycb", yp, ym, yc, yk, n_cls=22)
for name, fn in (("graph", lambda: gy(yp, ym, yc, yk)), ("polled", lambda: ev.cal_batch_poses(yp, ym, yc, yk, True, 22, True))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(20):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    print(name, "median ms/call %.3f" % (1e3 * float(np.median(t))), "fallbacks", gy.fallbacks)
# bare replay without the input copies and the flag read
torch.cuda.synchronize()
t = []
for _ in range(20):
    t0 = time.perf_counter(); gy.graph.replay(); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
print("bare replay median ms %.3f" % (1e3 * float(np.median(t))))
