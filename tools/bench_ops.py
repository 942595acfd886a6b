#!/usr/bin/env python3
"""Per-op timing at the bench shapes (F frames), for kernel optimisation A/B runs.
usage: python tools/bench_ops.py [--frames 64] [--ops group,interp,bq,nn,fps,ms] [--reps 5]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import SA_LEVELS, FP_LEVELS, make_inputs  # noqa: E402


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
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--ops", default="copy,group,interp,bq,nn,fps,ms")
    ap.add_argument("--n-obj", type=int, default=3072)
    args = ap.parse_args()
    ops = set(args.ops.split(","))
    dev = torch.device("cuda:0")
    from pvn3d_amd.lib.pointnet2_utils import _ext
    F = args.frames
    inp = make_inputs(F, 12288, args.n_obj, dev, 0)
    if "copy" in ops:
        a = torch.empty(256 * 1024 * 1024, device=dev)   # 1 GiB
        b = torch.empty_like(a)
        mn, md = timeit(lambda: b.copy_(a), args.reps)
        print("copy 1GiB->1GiB  %.1f us  %.2f TB/s (r+w)" % (mn * 1e3, 2 * a.numel() * 4 / mn / 1e9))
        mn, md = timeit(lambda: b.fill_(1.0), args.reps)
        print("fill 1GiB        %.1f us  %.2f TB/s (w)" % (mn * 1e3, a.numel() * 4 / mn / 1e9))
        del a, b
    # build per-level geometry once
    xyz = inp["pcld"]
    feats = inp["feats"]
    levels = []
    for li, (n_in, m, c, radii, nss, co) in enumerate(SA_LEVELS):
        sel = _ext.furthest_point_sampling(xyz, m)
        new_xyz = _ext.gather_points(xyz.transpose(1, 2).contiguous(), sel).transpose(1, 2).contiguous()
        i0, i1 = _ext.ball_query_pair(new_xyz, xyz, radii[0], nss[0], radii[1], nss[1])
        levels.append(dict(xyz=xyz, new_xyz=new_xyz, feats=feats, idx=(i0, i1), radii=radii, nss=nss, c=c, m=m, n=n_in))
        xyz, feats = new_xyz, inp["sa_feats"][li]
    tot = {}
    for li, lv in enumerate(levels):
        if "fps" in ops:
            mn, _ = timeit(lambda: _ext.furthest_point_sampling(lv["xyz"], lv["m"]), args.reps)
            print("fps   L%d n=%5d m=%4d            %8.1f us  (%.3f us/round)" % (li, lv["n"], lv["m"], mn * 1e3, mn * 1e3 / lv["m"]))
            tot["fps"] = tot.get("fps", 0) + mn
        if "bq" in ops:
            mn, _ = timeit(lambda: _ext.ball_query_pair(lv["new_xyz"], lv["xyz"], lv["radii"][0], lv["nss"][0], lv["radii"][1], lv["nss"][1]), args.reps)
            pairs = F * lv["n"] * lv["m"]
            print("bq    L%d n=%5d m=%4d            %8.1f us  %.2f Tpair/s" % (li, lv["n"], lv["m"], mn * 1e3, pairs / mn / 1e9))
            tot["bq"] = tot.get("bq", 0) + mn
        if "group" in ops:
            for s in range(2):
                idx = lv["idx"][s]
                mn, _ = timeit(lambda: _ext.group_xyz_features(lv["xyz"], lv["new_xyz"], lv["feats"], idx, True), args.reps)
                by = F * (3 + lv["c"]) * lv["m"] * lv["nss"][s] * 4
                print("group L%d C=%3d m=%4d ns=%2d        %8.1f us  %.2f TB/s (out bytes)" % (li, lv["c"], lv["m"], lv["nss"][s], mn * 1e3, by / mn / 1e9))
                tot["group"] = tot.get("group", 0) + mn
    l_xyz = [inp["pcld"]] + [lv["new_xyz"] for lv in levels]
    for fi, (n, mm, c) in enumerate(FP_LEVELS):
        unknown, known = l_xyz[3 - fi], l_xyz[4 - fi]
        d2, idx = _ext.three_nn(unknown, known)
        if "nn" in ops:
            mn, _ = timeit(lambda: _ext.three_nn(unknown, known), args.reps)
            print("nn    n=%5d m=%4d                %8.1f us  %.2f Tpair/s" % (n, mm, mn * 1e3, F * n * mm / mn / 1e9))
            tot["nn"] = tot.get("nn", 0) + mn
        if "interp" in ops:
            w = torch.rand(F, n, 3, device=dev)
            mn, _ = timeit(lambda: _ext.three_interpolate(inp["fp_known"][fi], idx, w), args.reps)
            print("interp C=%4d m=%4d n=%5d        %8.1f us  %.2f TB/s (out bytes)" % (c, mm, n, mn * 1e3, F * c * n * 4 / mn / 1e9))
            tot["interp"] = tot.get("interp", 0) + mn
    if "ms" in ops:
        from pvn3d_amd.lib.utils import pvn3d_eval_utils as ev
        for poll in (4, 0):
            mn, _ = timeit(lambda: ev.cal_batch_poses_lm(inp["pcld"], inp["mask"], inp["ctr_of"], inp["pred_kp_of"], True, 2, False, 1, poll_every=poll), args.reps)
            print("vote+cluster+pose F=%d n_obj=%d poll=%d  %8.1f us" % (F, args.n_obj, poll, mn * 1e3))
        tot["ms"] = mn
    if "msg" in ops:
        from pvn3d_amd.lib.pointnet2_msg import Pointnet2MSG
        from pvn3d_amd.lib.pointnet2_utils import pointnet2_modules as pm
        net = Pointnet2MSG(input_channels=6).to(dev).eval()
        pc = torch.cat([inp["pcld"], inp["feats"].transpose(1, 2)], 2).contiguous()
        with torch.no_grad():
            mn, _ = timeit(lambda: net(pc), args.reps)
            print("Pointnet2MSG fused   F=%d  %8.1f us  (%.1f us/frame)" % (F, mn * 1e3, mn * 1e3 / F))
            pm.FUSED_INFERENCE = False
            sub = pc[: min(F, 16)].contiguous()
            mn2, _ = timeit(lambda: net(sub), args.reps)
            pm.FUSED_INFERENCE = True
            print("Pointnet2MSG unfused F=%d  %8.1f us  (%.1f us/frame)" % (sub.size(0), mn2 * 1e3, mn2 * 1e3 / sub.size(0)))
        tot["msg"] = mn
    print("totals (ms):", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
