#!/usr/bin/env python3
"""Cycle stamps of the LDS-DMA split GEMM (tuning build: bash tools/sg_variants.sh 8; run with
PVN3D_HIP_LIB=tools/sgv/libsg_8.so): mean cycles per wave in each phase of a launch; wave lifetimes on the 100 MHz
real-time counter, occupancy per CU (HW_ID / XCC_ID), and the shader clock the two counters imply.
usage: PVN3D_HIP_LIB=tools/sgv/libsg_8.so python tools/sg_prof.py"""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd._lib import lib, check  # noqa: E402
from sg_time import LAUNCHES  # noqa: E402

PH = ("prologue + absmax", "vmcnt wait", "barrier", "DMA issue", "reads + MFMA", "epilogue: scales", "gather + bias + relu", "stores")


def main():
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rd = lib.pvn3d_sg_prof_read
    rd.restype = ctypes.c_int
    rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for name, P, N, K, zm, hout in LAUNCHES:
        S = (K + 31) // 32 * 2
        NP = (N + 127) // 128 * 128
        x = torch.randn(P, K, device=dev)
        w = torch.randn(NP, K, device=dev) / K ** 0.5
        xb, wb = x.abs().max().reshape(1).clone(), w.abs().max().reshape(1).clone()
        xs = torch.empty(P * S * 64, dtype=torch.uint8, device=dev)
        ws = torch.empty(NP * S * 64, dtype=torch.uint8, device=dev)
        check(lib.pvn3d_split_rows2(P, K, x.data_ptr(), K, xb.data_ptr(), xs.data_ptr(), S, st), "x")
        check(lib.pvn3d_split_rows2(NP, K, w.data_ptr(), K, wb.data_ptr(), ws.data_ptr(), S, st), "w")
        w_scale = 2.0 ** (14 - math.frexp(float(wb))[1])
        rm = torch.ones(NP, device=dev)
        bias = torch.randn(NP, device=dev)
        n_per = P // 64 if P % 64 == 0 else P
        z = idx = wgt = None
        if zm:
            z = torch.randn((P + n_per - 1) // n_per * zm, NP, device=dev)
            idx = torch.randint(0, zm, (P, 3), device=dev, dtype=torch.int32)
            wgt = torch.rand(P, 3, device=dev)
        out = torch.zeros(P, N, device=dev)
        S_out = NP // 16
        oh = torch.zeros(P * S_out * 64, dtype=torch.uint8, device=dev) if hout else None
        ob = torch.full((1,), 64.0, device=dev)
        am = torch.zeros(1, device=dev)

        def run():
            check(lib.pvn3d_split_gemm2(P, N, S, xs.data_ptr(), xb.data_ptr(), ws.data_ptr(), w_scale, rm.data_ptr(),
                                        bias.data_ptr(), 1, z.data_ptr() if zm else None, NP, n_per, zm,
                                        idx.data_ptr() if zm else None, wgt.data_ptr() if zm else None,
                                        None if hout else out.data_ptr(), N, None if hout else am.data_ptr(), oh.data_ptr() if hout else None, S_out,
                                        ob.data_ptr() if hout else None, st), "gemm")
        run(); run()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        wide = (NP // 128) * ((P + 255) // 256) >= 512
        waves = (NP // 128) * ((P + (255 if wide else 127)) // (256 if wide else 128)) * 4
        rd(buf, waves)
        life = (ctypes.c_uint * (4 * waves))()
        lf = lib.pvn3d_sg_prof_life
        lf.restype = ctypes.c_int
        lf.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lf(life, waves)
        import numpy as np
        L = np.frombuffer(life, dtype=np.uint32).reshape(-1, 4).astype(np.int64)
        t0 = L[:, 0].min()
        start, end = (L[:, 0] - t0) / 100.0, (L[:, 1] - t0) / 100.0              # us
        cu = (L[:, 3] & 15) * 4096 + ((L[:, 2] >> 13) & 7) * 256 + ((L[:, 2] >> 12) & 1) * 64 + ((L[:, 2] >> 8) & 15)
        simd = (L[:, 2] >> 4) & 3
        per_cu = {}
        for c, a_, b_ in zip(cu, start, end):
            per_cu.setdefault(int(c), []).append((a_, b_))
        conc = []
        for c, iv in per_cu.items():
            ev = sorted([(a_, 1) for a_, _ in iv] + [(b_, -1) for _, b_ in iv])
            cur = mx = 0
            for _, d in ev:
                cur += d
                mx = max(mx, cur)
            conc.append(mx)
        tot = sum(buf)
        print("   wave lifetimes: mean %.1f us, launch span %.1f us, last start %.1f us; CUs used %d, waves per CU %.1f, "
              "max concurrent waves per CU: min %d mean %.1f max %d; shader clock = cycle stamps / real-time stamps = %.2f GHz"
              % (float((end - start).mean()), float(end.max()), float(start.max()), len(per_cu), waves / len(per_cu),
                 min(conc), sum(conc) / len(conc), max(conc), tot / waves / max(1e-9, float((end - start).mean())) / 1e3))
        print("%-28s %7.1f us  %6d waves, %d stages, tile x %d points: cycles per wave %8.0f = " % (
            name, e0.elapsed_time(e1) * 1e3, waves, S, 256 if wide else 128, tot / waves) +
            "  ".join("%s %.0f" % (PH[i], buf[i] / waves) for i in range(8)) +
            "   | per stage: wait %.0f barrier %.0f issue %.0f mfma %.0f" % tuple(buf[i] / waves / S for i in (1, 2, 3, 4)),
            flush=True)


if __name__ == "__main__":
    main()
