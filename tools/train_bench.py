#!/usr/bin/env python3
"""Training-step timing of the voting branch (BASELINE config 5) on one GPU, for profiling:
python tools/train_bench.py [--frames 24] [--steps 5] [--dtype bf16|fp32]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd import train_step as ts  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--prefetch", action="store_true", help="enqueue the next step's geometry under this step's backward")
    ap.add_argument("--library-mlp", action="store_true",
                    help="SharedMLP through torch Conv2d / BatchNorm2d (MIOpen / hipBLASLt) instead of csrc/mlp_train.hip")
    args = ap.parse_args()
    if args.library_mlp:
        from pvn3d_amd.lib.pointnet2_utils import _train_mlp
        _train_mlp.TRAIN_FUSED = False
    dev = torch.device("cuda:0")
    batch = ts.synthetic_batch(args.frames, 12288, dev, seed_base=7500, n_obj=3072)
    torch.manual_seed(1)
    model = ts.PointVoteNet().to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    dt = torch.bfloat16 if args.dtype == "bf16" else None
    for _ in range(2):
        ts.train_step(model, opt, batch, autocast_dtype=dt, prefetch=batch["pc"] if args.prefetch else None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = ts.train_step(model, opt, batch, autocast_dtype=dt, prefetch=batch["pc"] if args.prefetch else None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    print("train_step %s: %.2f ms / step of %d frames = %.1f frames/s, loss %.3f" %
          (args.dtype, ms, args.frames, args.frames * 1e3 / ms, float(loss)))


if __name__ == "__main__":
    main()
