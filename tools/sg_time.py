#!/usr/bin/env python3
"""Split GEMM (csrc/split_gemm.hip), fp16 x 2: the LDS-DMA kernel (pvn3d_split_gemm2) against the register-staged
128 x 128-tile kernel (pvn3d_split_gemm2_tile128) at the launches of the 64-frame forward: time, MFMA rate (three partial
products per multiply against the 2.5 PFLOP/s fp16 pipe), and a bit comparison of the two results (the MFMA order per
accumulator is the same in both), repeated --reps times on fresh random operands to catch an ordering bug of the DMA ring.
usage: python tools/sg_time.py [--reps 3] [--only NAME]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd._lib import lib, check  # noqa: E402

# name, points, out channels, contraction, interpolated rows (known points per frame, 0 = none), h16 output
LAUNCHES = (
    ("FP3 Z   (Wa.known)", 64 * 128, 512, 1024, 0, False),
    ("FP3 H   (Wb.skip + interp)", 64 * 512, 512, 512, 128, True),
    ("FP3 Y", 64 * 512, 512, 512, 0, False),
    ("FP2 Z", 64 * 512, 512, 512, 0, False),
    ("FP2 H", 64 * 1024, 512, 256, 512, True),
    ("FP2 Y", 64 * 1024, 512, 512, 0, False),
    ("SA3 pre (2 scales)", 64 * 512, 512, 512, 0, False),
    ("SA2 pre (2 scales)", 64 * 1024, 256, 256, 0, False),
    ("FP1 pre", 64 * 1024, 256, 512, 0, False),
    ("FP0 pre", 64 * 2048, 128, 256, 0, False),
    ("ragged", 64 * 512 - 77, 200, 96, 128, True),
)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    tot = {"dma": 0.0, "t128": 0.0}
    for name, P, N, K, zm, hout in LAUNCHES:
        if args.only and args.only not in name:
            continue
        S = (K + 31) // 32 * 2
        NP = (N + 127) // 128 * 128
        worst_same = True
        times = {}
        for rep in range(args.reps):
            g = torch.Generator(device=dev).manual_seed(1000 * rep + P % 997)
            x = torch.randn(P, K, device=dev, generator=g)
            w = torch.randn(NP, K, device=dev, generator=g) / K ** 0.5
            w[N:] = 0
            xb = x.abs().max().reshape(1).clone()
            wb = w.abs().max().reshape(1).clone()
            xs = torch.empty(P * S * 64, dtype=torch.uint8, device=dev)
            ws = torch.empty(NP * S * 64, dtype=torch.uint8, device=dev)
            check(lib.pvn3d_split_rows2(P, K, x.data_ptr(), K, xb.data_ptr(), xs.data_ptr(), S, st), "split x")
            # the weights as the host packs them: a power-of-two scale that puts max|w| at 2^14 -> here through the same kernel
            check(lib.pvn3d_split_rows2(NP, K, w.data_ptr(), K, wb.data_ptr(), ws.data_ptr(), S, st), "split w")
            import math
            w_scale = 2.0 ** (14 - math.frexp(float(wb))[1])
            rm = (2.0 ** torch.randint(-3, 4, (NP,), device=dev)).float()
            bias = torch.randn(NP, device=dev)
            bias[N:] = 0
            z = idx = wgt = None
            n_per = P // 64 if P % 64 == 0 else P
            if zm:
                frames = (P + n_per - 1) // n_per
                z = torch.randn(frames * zm, NP, device=dev, generator=g)
                z[:, N:] = 0
                idx = torch.randint(0, zm, (P, 3), device=dev, dtype=torch.int32)
                wgt = torch.rand(P, 3, device=dev)
                wgt = wgt / wgt.sum(1, keepdim=True)
            S_out = NP // 16
            ob = torch.full((1,), 64.0, device=dev)
            outs = {}
            for key, fn in (("dma", lib.pvn3d_split_gemm2), ("t128", lib.pvn3d_split_gemm2_tile128)):
                out = torch.zeros(P, N, device=dev)
                oh = torch.zeros(P * S_out * 64, dtype=torch.uint8, device=dev) if hout else None
                am = torch.zeros(1, device=dev)

                def run():
                    check(fn(P, N, S, xs.data_ptr(), xb.data_ptr(), ws.data_ptr(), w_scale, rm.data_ptr(), bias.data_ptr(), 1,
                             z.data_ptr() if zm else None, NP, n_per, zm, idx.data_ptr() if zm else None,
                             wgt.data_ptr() if zm else None, None if hout else out.data_ptr(), N, None if hout else am.data_ptr(),
                             oh.data_ptr() if hout else None, S_out, ob.data_ptr() if hout else None, st), key)
                run()
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run()
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                times[key] = min(min(ts), times.get(key, 1e30))
                outs[key] = (out.clone(), oh.clone() if hout else None, am.clone())
            same = torch.equal(outs["dma"][0], outs["t128"][0]) and torch.equal(outs["dma"][2], outs["t128"][2]) and \
                (not hout or torch.equal(outs["dma"][1], outs["t128"][1]))
            worst_same = worst_same and bool(same)
            if rep == 0:
                # against fp64 of the same operands (per-row multipliers folded in)
                ref = (x.double() @ (w[:N].double() * rm[:N].double()[:, None]).T)
                if zm:
                    f = torch.arange(P, device=dev) // n_per
                    rows = z.double()[(f[:, None] * zm + idx.long())]            # (P, 3, NP)
                    ref = ref + (rows[:, :, :N] * wgt.double()[:, :, None]).sum(1)
                ref = torch.relu(ref + bias[:N].double())
                err = float("nan") if hout else float((outs["dma"][0].double() - ref).abs().max() / ref.abs().max())
        flops = 2.0 * P * NP * (S * 16) * 3
        for k in tot:
            tot[k] += times[k]
        print("%-28s P=%6d N=%3d K=%4d  LDS-DMA %7.1f us (%5.0f TF/s, %.2f of the fp16 pipe)   tile128 %7.1f us (%.2f)   "
              "bits equal in %d runs: %s   max err vs fp64 / scale %.1e"
              % (name, P, N, K, times["dma"], flops / times["dma"] / 1e6, flops / times["dma"] / 1e6 / 2500.0, times["t128"],
                 flops / times["t128"] / 1e6 / 2500.0, args.reps, worst_same, err), flush=True)
    print("sum: LDS-DMA %.1f us, tile128 %.1f us" % (tot["dma"], tot["t128"]))


if __name__ == "__main__":
    main()
