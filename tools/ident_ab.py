#!/usr/bin/env python3
"""PVN3D_MLP_IDENTITY_A A/B: a 64-frame Pointnet2MSG forward with and without the flag (pre-contracted chains: the
(weight low piece x activation high piece) product of the identity block not issued) -- same output bits, time per
forward and per stage."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from pvn3d_amd.lib.pointnet2_utils import _ext  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    net = bench.make_net(dev)
    inp = bench.make_inputs(64, 12288, 3072, dev, seed_base=4300)
    pc = torch.cat([inp["pcld"], inp["feats"].transpose(1, 2)], 2).contiguous()
    out = {}
    for rep in range(3):
        for flag in (True, False):
            _ext.IDENTITY_SKIP = flag
            with torch.no_grad():
                o = net(pc)
                d = hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:16]
                ms = bench._median_ms(lambda: net(pc), 7)
            out.setdefault(flag, []).append((d, ms))
    _ext.IDENTITY_SKIP = True
    for flag in (True, False):
        print("identity skip %-5s: digests %s   forward ms %s" % (flag, sorted(set(d for d, _ in out[flag])),
                                                                  ["%.3f" % m for _, m in out[flag]]))
    print("same bits:", set(d for d, _ in out[True]) == set(d for d, _ in out[False]))


if __name__ == "__main__":
    main()
