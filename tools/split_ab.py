#!/usr/bin/env python3
"""Digest of (i) pvn3d_split_rows2 over adversarial values and (ii) a Pointnet2MSG forward + the split GEMM launches,
for an A/B of two builds of the library (PVN3D_HIP_LIB): `python tools/split_ab.py` with the shipped library and with
tools/ab/libplain.so (bash tools/split_ab.sh) must print the same digests -- the v_fma_mix form of the fp16 x 2 split
(csrc/common.h) returns the bits of the straightforward form.  Also prints the forward's time."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd._lib import lib, check, LIB_PATH  # noqa: E402
import bench  # noqa: E402


def digest(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


def main():
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(5)
    rows, K = 4096, 256
    x = torch.randn(rows, K, generator=g) * torch.pow(10.0, torch.rand(rows, K, generator=g) * 12 - 8)
    # halfway cases of the fp16 grid, exact fp16 values, powers of two, zeros, the smallest magnitudes
    h = torch.randn(rows, 16, generator=g).half().float()
    x[:, :16] = h
    x[:, 16:32] = h * (1 + 2.0 ** -11)
    x[:, 32:48] = h * (1 + 2.0 ** -12)
    x[:, 48:64] = torch.pow(2.0, torch.randint(-40, 14, (rows, 16), generator=g).float())
    x[:, 64:72] = 0.0
    x[:, 72:80] = -0.0
    x[:, 80:96] = torch.randn(rows, 16, generator=g) * 1e-30
    x = x.to(dev)
    S = K // 16
    for bound in (float(x.abs().max()), 1.0, 1e-6, 3e7):
        b = torch.tensor([bound], device=dev)
        out = torch.zeros(rows * S * 64, dtype=torch.uint8, device=dev)
        check(lib.pvn3d_split_rows2(rows, K, x.data_ptr(), K, b.data_ptr(), out.data_ptr(), S, st), "split_rows2")
        print("split_rows2 bound %.3g: %s" % (bound, digest(out)))
    net = bench.make_net(dev)
    inp = bench.make_inputs(16, 12288, 3072, dev, seed_base=4200)
    pc = torch.cat([inp["pcld"], inp["feats"].transpose(1, 2)], 2).contiguous()
    with torch.no_grad():
        o = net(pc)
        print("Pointnet2MSG forward, 16 frames: %s" % digest(o))
        ms = bench._median_ms(lambda: net(pc), 10)
    print("forward %.3f ms   (library: %s)" % (ms, LIB_PATH))


if __name__ == "__main__":
    main()
