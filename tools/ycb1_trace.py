#!/usr/bin/env python3
"""One YCB multi-instance frame per call through cal_batch_poses (for rocprofv3 --kernel-trace) + host-side timing."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd import synth
from pvn3d_amd.lib.utils import pvn3d_eval_utils as ev
dev = torch.device("cuda:0")
fy = [synth.synth_frame_ycb(frame=7200)]
sty = lambda k: torch.from_numpy(np.stack([f[k] for f in fy], 0)).to(dev)
yp, ym, yc, yk = sty("pcld").contiguous(), sty("mask").to(torch.int32).contiguous(), sty("ctr_of").contiguous(), sty("pred_kp_of").contiguous()
poll = int(sys.argv[1]) if len(sys.argv) > 1 else 4
run = lambda: ev.cal_batch_poses(yp, ym, yc, yk, True, 22, True, poll_every=poll)
for _ in range(5):
    run()
torch.cuda.synchronize()
ts, hs = [], []
for _ in range(20):
    t0 = time.perf_counter(); r = run(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append((t1 - t0) * 1e3); ts.append((t2 - t0) * 1e3)
print("poll_every %d: call returns after %.3f ms (host), done after %.3f ms" % (poll, np.median(hs), np.median(ts)))
