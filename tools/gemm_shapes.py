#!/usr/bin/env python3
"""Time csrc/mlp_train.hip's bf16 GEMM (C = A.B^T with the BatchNorm-statistics epilogue) at the shapes of the 24-frame
training step and print the effective HBM rate (A read + C written)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd._lib import lib, check
dev = torch.device("cuda:0")
F = 24
shapes = []
for name, rows, chans in (("SA0s1", F * 2048 * 16, (9, 16, 16, 32)), ("SA0s2", F * 2048 * 32, (9, 32, 32, 64)),
                          ("SA1s1", F * 1024 * 16, (99, 64, 64, 128)), ("SA1s2", F * 1024 * 32, (99, 64, 96, 128)),
                          ("SA2s1", F * 512 * 16, (259, 128, 196, 256)), ("SA2s2", F * 512 * 32, (259, 128, 196, 256)),
                          ("SA3s1", F * 128 * 16, (515, 256, 256, 512)), ("SA3s2", F * 128 * 32, (515, 256, 384, 512)),
                          ("FP0", F * 12288, (262, 128, 128, 128))):
    for i in range(len(chans) - 1):
        shapes.append(("%s.l%d" % (name, i), rows, chans[i], chans[i + 1]))
ld = lambda c: (c + 15) // 16 * 16
st = torch.cuda.current_stream().cuda_stream
tot_t = tot_b = 0.0
for name, M, K, N in shapes:
    K_, N_ = ld(K), ld(N)
    A = torch.randn((M, K_), device=dev).to(torch.bfloat16)
    B = torch.randn((N, K_), device=dev).to(torch.bfloat16)
    C = torch.empty((M, N_), dtype=torch.bfloat16, device=dev)
    P = lib.pvn3d_mt_gemm_nt_stat_rows(M)
    ss = torch.empty((2, P, N_), dtype=torch.float32, device=dev)
    run = lambda: check(lib.pvn3d_mt_gemm_nt(M, N, K_, A.data_ptr(), K_, B.data_ptr(), K_, C.data_ptr(), N_, ss[0].data_ptr(),
                                             ss[1].data_ptr(), N_, st), "gemm")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 100
    byts = M * (K_ + N_) * 2
    tot_t += us; tot_b += byts
    print("%-9s M=%8d K=%4d N=%4d  %7.1f us  %5.2f TB/s  %6.1f TFLOP/s" % (name, M, K_, N_, us, byts / us / 1e6, 2.0 * M * K_ * N_ / us / 1e6), flush=True)
print("forward GEMMs of one step: %.2f ms, %.2f GB, %.2f TB/s" % (tot_t / 1e3, tot_b / 1e9, tot_b / tot_t / 1e6))
