#!/usr/bin/env python3
"""Bisect: why do iterations 2-4 on the headline's real votes take 1.2 ms when synthetic votes of the same statistics take 0.97 ms?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import make_inputs, run_postproc, StageTimer
from pvn3d_amd.lib.utils import _vote_engine as eng
dev = torch.device("cuda:0")
inp = make_inputs(64, 12288, 3072, dev, 0)
cap = []
real = eng.meanshift_fit_batch
def spy(*a, **k):
    cap.append((a, dict(k))); return real(*a, **k)
eng.meanshift_fit_batch = spy
run_postproc(inp, StageTimer(False), 4)
torch.cuda.synchronize()
eng.meanshift_fit_batch = real
(pts, off, cnt, max_cnt, bw), k = cap[0][0][:5], cap[0][1]
def rate(pts, off, cnt, max_cnt, label):
    kk = "sgpr+nowin+noearly"
    f = lambda lim: real(pts, off, cnt, max_cnt, bw, max_iter=300, aligned32=True, kernel=kk, enqueue_limit=lim)
    f(4); torch.cuda.synchronize()
    def t(lim):
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = f(lim); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        return float(np.median(ts)), r
    (t4, r4), (t1, _) = t(4), t(1)
    print("%-58s iterations 2-4: %.3f ms each   (iters min %d max %d)" % (label, (t4 - t1) / 3, int(r4[2].min()), int(r4[2].max())), flush=True)
rate(pts, off, cnt, max_cnt, "real votes, real layout")
n_seg, stride = off.numel(), int(off[1].item())
P = pts.view(n_seg, stride, 4)
g = torch.Generator().manual_seed(1)
sig = torch.where(torch.rand(n_seg, 3072, 1, generator=g) < 0.1, 0.05, 0.005)
v = (torch.randn(n_seg, 3072, 3, generator=g) * sig + torch.tensor([0.05, -0.02, 0.9])).to(dev)
Q = torch.zeros_like(P); Q[:, :3072, :3] = v
rate(Q.view(-1, 4), off, cnt, max_cnt, "synthetic votes in the real layout (same off / cnt tensors)")
Q2 = P.clone(); Q2[:, 3072:] = 0
rate(Q2.view(-1, 4), off, cnt, max_cnt, "real votes, rows beyond the count zeroed")
perm = torch.randperm(n_seg, generator=g).to(dev)
rate(P[perm].contiguous().view(-1, 4), off, cnt, max_cnt, "real votes, segments shuffled")
Q3 = P.clone(); Q3[:, :3072, :3] = P[:, :3072, :3][:, torch.randperm(3072, generator=g).to(dev)]
rate(Q3.view(-1, 4), off, cnt, max_cnt, "real votes, rows shuffled inside every segment")
c = P[:, :3072, :3].mean(1, keepdim=True)
Q4 = P.clone(); Q4[:, :3072, :3] = P[:, :3072, :3] - c + torch.tensor([0.05, -0.02, 0.9], device=dev)
rate(Q4.view(-1, 4), off, cnt, max_cnt, "real votes, every segment re-centred on one point")
for mc in (12288, 12320, 12416, 12544, 3072, 3200, 6144, 6272):
    rate(pts, off, cnt, mc, "real votes, real layout, max_cnt_host = %d (%d tiles per fit)" % (mc, (mc + 127) // 128))
