#!/usr/bin/env python3
"""vote -> cluster -> pose timing on the heavy-tailed vote set of bench.py (16 frames, 10 % outliers of sigma 30 cm)
and on the headline set: python tools/ms_heavy_time.py [--reps 5] [--sets heavy,headline]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd import synth  # noqa: E402
from pvn3d_amd.lib.utils import pvn3d_eval_utils as ev  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--sets", default="heavy,headline")
    ap.add_argument("--polls", default="4,0")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for name in args.sets.split(","):
        kw, nf = (dict(sig_out=0.30), 16) if name == "heavy" else (dict(), 64)
        fh = [synth.synth_frame(frame=7400 + i, n_pts=12288, n_obj=3072, **kw) for i in range(nf)]
        st = lambda k: torch.from_numpy(np.stack([f[k] for f in fh], 0)).to(dev)
        a = (st("pcld").contiguous(), st("mask").to(torch.int32).contiguous(), st("ctr_of").contiguous(),
             st("pred_kp_of").contiguous(), True, 2, False, 1)
        for pe in (int(v) for v in args.polls.split(",")):
            for _ in range(2):
                r = ev.cal_batch_poses_lm(*a, poll_every=pe)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                r = ev.cal_batch_poses_lm(*a, poll_every=pe)
            torch.cuda.synchronize()
            print("%s poll_every %d: %.3f ms/frame, iters max %d" %
                  (name, pe, (time.perf_counter() - t0) / args.reps / nf * 1e3, int(r["iters"].max())), flush=True)


if __name__ == "__main__":
    main()
