#!/usr/bin/env python3
"""FPS level-0 timing (12288 -> 2048): culled kernel with one wave per 64-point slot (default) / with one wave per cloud,
and the register-resident kernel, at F frames.
usage: python tools/fps_time.py [--frames 1,64] [--reps 5]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd import synth  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="1,64")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--shapes", default="12288:2048,8192:2048,12288:512")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    from pvn3d_amd.lib.pointnet2_utils import _ext
    for shape in args.shapes.split(","):
        n, m = (int(v) for v in shape.split(":"))
        for F in (int(v) for v in args.frames.split(",")):
            xyz = torch.from_numpy(np.stack([synth.synth_cloud(np.random.default_rng(1234 + f), n)[0]
                                             for f in range(F)], 0)).to(dev)
            out = {}
            for key, culled, waves in (("mw", True, 3), ("one", True, 0), ("reg", False, 0)):
                _ext.FPS_CULLED, _ext.FPS_WAVES = culled, waves
                mn, md = timeit(lambda: _ext.furthest_point_sampling(xyz, m), args.reps)
                out[key] = (mn, _ext.furthest_point_sampling(xyz, m))
            _ext.FPS_CULLED, _ext.FPS_WAVES = True, 0
            same = bool(torch.equal(out["mw"][1], out["reg"][1])) and bool(torch.equal(out["one"][1], out["reg"][1]))
            print("fps n=%5d m=%4d F=%3d  culled, one wave per slot %8.1f us (%.3f us/round)   culled, one wave %8.1f us "
                  "(%.3f us/round)   register-resident %8.1f us   same=%s"
                  % (n, m, F, out["mw"][0] * 1e3, out["mw"][0] * 1e3 / m, out["one"][0] * 1e3, out["one"][0] * 1e3 / m,
                     out["reg"][0] * 1e3, same), flush=True)


if __name__ == "__main__":
    main()
