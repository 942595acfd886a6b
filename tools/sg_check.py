#!/usr/bin/env python
"""csrc/split_gemm.hip against float64 torch: plain GEMM (fp32 out, s16 out), the gathered-add epilogue."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvn3d_amd._lib import lib, check
from pvn3d_amd.lib.pointnet2_utils import _fused_mlp as fm

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)


def s16_to_float(buf, rows, S):
    v = buf.view(torch.int16).view(rows, S, 3, 16).to(torch.int32) << 16
    return v.view(torch.float32).double().sum(2).reshape(rows, S * 16)


for (P, K, N) in ((300, 70, 200), (1024, 512, 512), (4096, 1536, 512)):
    X = torch.randn(P, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    S = fm._slabs(K)
    xs = torch.empty(P * S * 96, dtype=torch.uint8, device=dev)
    check(lib.pvn3d_split_rows(P, K, X.data_ptr(), K if K % 4 == 0 else K, xs.data_ptr(), S, st), "split_rows") if K % 4 == 0 else None
    if K % 4:
        Xp = torch.zeros(P, (K + 3) // 4 * 4, device=dev); Xp[:, :K] = X
        check(lib.pvn3d_split_rows(P, K, Xp.data_ptr(), Xp.size(1), xs.data_ptr(), S, st), "split_rows")
    back = s16_to_float(xs, P, S)[:, :K]
    print("split_rows exact:", bool((back == X.double()).all()))
    ws = fm._pack_weight_s16(W, S)
    Np = ws.size(0)
    bp = torch.zeros(Np, device=dev); bp[:N] = b
    out = torch.full((P, Np), float("nan"), device=dev)
    Sout = fm._slabs(N)
    outs = torch.empty(P * Sout * 96, dtype=torch.uint8, device=dev)
    check(lib.pvn3d_split_gemm(P, N, S, xs.data_ptr(), ws.data_ptr(), bp.data_ptr(), 1, None, 0, 0, 0, None, None,
                               out.data_ptr(), Np, outs.data_ptr(), Sout, st), "gemm")
    want = torch.relu(X.double() @ W.double().T + b.double())
    got = out[:, :N].double()
    print((P, K, N), "fp32 out err", float((got - want).abs().max()), "s16 out == fp32 out:",
          bool((s16_to_float(outs, P, Sout)[:, :N] == got).all()), "pad zero:", bool((s16_to_float(outs, P, Sout)[:, N:] == 0).all()))
    # gathered add
    B, n, m = 4, P // 4, 37
    Z = torch.randn(B * m, Np, device=dev)
    idx = torch.randint(0, m, (P, 3), device=dev, dtype=torch.int32)
    wg = torch.rand(P, 3, device=dev)
    out2 = torch.empty((P, Np), device=dev)
    check(lib.pvn3d_split_gemm(B * n, N, S, xs.data_ptr(), ws.data_ptr(), bp.data_ptr(), 1, Z.data_ptr(), Np, n, m,
                               idx.data_ptr(), wg.data_ptr(), out2.data_ptr(), Np, None, 0, st), "gemm z")
    f = (torch.arange(B * n, device=dev) // n).long()
    zg = sum(Z.double()[f * m + idx[:B * n, t].long()] * wg[:B * n, t:t + 1].double() for t in range(3))
    want2 = torch.relu(X[:B * n].double() @ W.double().T + zg[:, :N] + b.double())
    print("   gathered add err", float((out2[:B * n, :N].double() - want2).abs().max()))
