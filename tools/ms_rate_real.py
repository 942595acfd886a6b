#!/usr/bin/env python3
"""The iteration launches of the headline vote batch on ITS OWN votes (captured from one run_postproc call), timed like
tools/ms_rate.py: per-iteration time with the sgpr / LDS kernels, and what the votes look like in the kernels' scaled units."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import make_inputs, run_postproc, StageTimer
from pvn3d_amd.lib.utils import _vote_engine as eng
dev = torch.device("cuda:0")
inp = make_inputs(64, 12288, 3072, dev, 0)
cap = []
real = eng.meanshift_fit_batch
def spy(*a, **k):
    cap.append((a, dict(k)))
    return real(*a, **k)
eng.meanshift_fit_batch = spy
import pvn3d_amd.lib.utils.pvn3d_eval_utils as ev
for mod in (ev,):
    if hasattr(mod, "meanshift_fit_batch"):
        mod.meanshift_fit_batch = spy
run_postproc(inp, StageTimer(False), 4)
torch.cuda.synchronize()
eng.meanshift_fit_batch = real
print("captured calls:", [(tuple(x.shape) if torch.is_tensor(x) else x for x in a[:4]) for a, k in cap][:0], len(cap))
for a, k in cap:
    pts, off, cnt, max_cnt, bw = a[:5]
    print("call: pts", tuple(pts.shape), "segments", off.numel(), "max_cnt", max_cnt, "bw", bw, {kk: vv for kk, vv in k.items() if kk != "labels"})
    c = cnt.cpu().numpy(); o = off.cpu().numpy()
    P = pts.cpu().numpy()
    kappa = np.sqrt(0.5 * np.log2(np.e)) / bw
    seg = 5
    v = P[o[seg]:o[seg] + c[seg], :3]
    rel = (v - v[0]) * kappa
    print("  segment %d: count %d, |c'|^2 max %.1f (fast form needs <= 64), spread (std, m) %s" % (seg, c[seg], (rel ** 2).sum(1).max(), v.std(0)))
    for kern in ("sgpr", "packed+split"):
        kk = kern + "+nowin+noearly"
        def f(lim):
            return real(pts, off, cnt, max_cnt, bw, max_iter=300, aligned32=k.get("aligned32", False), kernel=kk, enqueue_limit=lim)
        f(6); torch.cuda.synchronize()
        def t(lim):
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(lim); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
            return float(np.median(ts))
        t6, t1 = t(6), t(1)
        per = (t6 - t1) / 5
        pairs = float((c.astype(np.float64) ** 2).sum())
        print("  %-14s %.3f ms per iteration = %.2fe12 pairs/s (1 iteration + everything else: %.3f ms)" % (kern, per, pairs / per / 1e9, t1))
