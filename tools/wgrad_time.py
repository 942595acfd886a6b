#!/usr/bin/env python3
"""mt_wgrad_tn at the narrow-layer shapes of the 24-frame training step."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd._lib import lib, check
dev = torch.device("cuda:0")
F = 24
st = torch.cuda.current_stream().cuda_stream
ld = lambda c: (c + 15) // 16 * 16
tot = 0.0
for name, rows, M, N in (("SA0s2.l2", F * 2048 * 32, 64, 32), ("SA0s2.l1", F * 2048 * 32, 32, 32), ("SA1s1.l2", F * 1024 * 16, 128, 64),
                         ("SA1s2.l0", F * 1024 * 32, 64, 99), ("SA1s2.l2", F * 1024 * 32, 128, 96), ("FP0.l1", F * 12288, 128, 128)):
    dY = torch.randn((rows, ld(M)), device=dev).to(torch.bfloat16)
    H = torch.randn((rows, ld(N)), device=dev).to(torch.bfloat16)
    dW = torch.zeros((M, N), device=dev)
    run = lambda: check(lib.pvn3d_mt_wgrad_tn(rows, M, N, dY.data_ptr(), ld(M), H.data_ptr(), ld(N), dW.data_ptr(), N, st), "w")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 100
    byts = rows * (ld(M) + ld(N)) * 2
    tot += us
    print("%-9s rows=%8d M=%3d N=%3d  %7.1f us  %5.2f TB/s (minimal reads)" % (name, rows, M, N, us, byts / us / 1e6), flush=True)
print("total %.0f us" % tot)
