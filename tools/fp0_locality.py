#!/usr/bin/env python3
"""FP level 0 (12288 <- 2048 points, [262 -> 128 -> 128], 64 frames): time of the fused module call with the real
three_nn neighbours (cloud order: consecutive points are spatially unrelated) against (a) neighbour lists that are local
in the table (point p -> rows p/6 + {0, 1, 2}) and (b) the real neighbours with the points processed in the order of their
nearest known point (rows permuted before the call, same arithmetic) -- what a locality-sorted tile order could buy."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd.lib.pointnet2_utils import pointnet2_modules as pm  # noqa: E402
from pvn3d_amd import synth  # noqa: E402


def med(fn, reps=9):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    B, n, m = 64, 12288, 2048
    torch.manual_seed(0)
    fp = pm.PointnetFPModule(mlp=[262, 128, 128]).to(dev).eval()
    unk = torch.from_numpy(np.stack([synth.synth_frame(frame=i, n_pts=n, n_obj=256)["pcld"] for i in range(B)])).to(dev)
    kn = unk[:, :m].contiguous()
    kf = torch.randn(B, m, 256, device=dev).transpose(1, 2)
    uf = torch.randn(B, n, 9, device=dev)[:, :, 3:].transpose(1, 2)
    with torch.no_grad():
        idx, w = fp.neighbours(unk, kn)
        t_real = med(lambda: fp(unk, kn, uf, kf, neighbours=(idx, w)))
        loc = (torch.arange(n, device=dev, dtype=torch.int32) // 6).clamp(max=m - 3)
        idx_l = torch.stack([loc, loc + 1, loc + 2], 1)[None].expand(B, n, 3).contiguous()
        t_loc = med(lambda: fp(unk, kn, uf, kf, neighbours=(idx_l, w)))
        order = torch.argsort(idx[..., 0].long(), dim=1, stable=True)                      # (B, n)
        g = lambda t: torch.gather(t, 1, order[..., None].expand(-1, -1, t.size(-1)))
        unk_s, idx_s, w_s = g(unk).contiguous(), g(idx).contiguous(), g(w).contiguous()
        uf_s = g(uf.transpose(1, 2).contiguous()).contiguous().transpose(1, 2)             # point-major view kept
        t_sort = med(lambda: fp(unk_s, kn, uf_s, kf, neighbours=(idx_s, w_s)))
    print("FP level 0, 64 frames: real neighbours %.3f ms   table-local neighbours %.3f ms   real neighbours, points in the "
          "order of their nearest known point %.3f ms" % (t_real, t_loc, t_sort))


if __name__ == "__main__":
    main()
