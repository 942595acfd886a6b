#!/usr/bin/env python3
"""Time the split-bf16 fused chains (csrc/sa_mlp_split.hip) of the backbone shapes against the fp32-MFMA kernels,
64 frames; PVN3D_S3_DBG (see sa_mlp_split.hip) switches pieces off for tuning.  Usage: python tools/s3_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd.lib.pointnet2_utils import pointnet2_modules as pm, _fused_mlp  # noqa: E402
from pvn3d_amd import synth  # noqa: E402
import numpy as np  # noqa: E402


def ms_of(fn, reps=9):
    """Median of `reps` single timed calls (a mean over a handful of calls showed 1.6 - 4.5 ms for a 0.3 ms kernel now and
    then: one host hiccup between the launches of a module call lands in the average)."""
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    B = 64
    torch.manual_seed(0)
    xyz = torch.from_numpy(np.stack([synth.synth_frame(frame=i, n_pts=1024, n_obj=256)["pcld"] for i in range(B)])).to(dev)
    cases = []
    # SA level 0: 12288 -> 2048 centres, 6 features read in place from the (B, N, 9) cloud, ns 16 / 32 (narrow-chain kernel)
    pc0 = torch.from_numpy(np.stack([np.concatenate([f["pcld"], f["feats"].T], 1).astype(np.float32) for f in
                                     (synth.synth_frame(frame=i, n_pts=12288, n_obj=3072) for i in range(B))])).to(dev)
    xyz0 = pc0[..., :3].contiguous()
    for ns, radius, mlp in ((16, 0.0175, [6, 16, 16, 32]), (32, 0.025, [6, 32, 32, 64])):
        sa = pm.PointnetSAModule(mlp=list(mlp), npoint=2048, radius=radius, nsample=ns).to(dev).eval()
        feats = pc0[..., 3:].transpose(1, 2)
        with torch.no_grad():
            geo = sa.sample_and_query(xyz0)
        cases.append(("SA0 ns%d" % ns, lambda sa=sa, feats=feats, geo=geo: sa(xyz0, feats, geometry=geo),
                      2.0 * (9 * mlp[1] + mlp[1] * mlp[2] + mlp[2] * mlp[3]) * 2048 * ns * B))
    # SA level 1: 2048 -> 1024 centres, C = 96, ns 16 / 32 (fp16 x 2 only: chain signature 111)
    xyz1 = torch.from_numpy(np.stack([synth.synth_frame(frame=i, n_pts=2048, n_obj=256)["pcld"] for i in range(B)])).to(dev)
    for ns, radius, mlp in ((16, 0.025, [96, 64, 64, 128]), (32, 0.05, [96, 64, 96, 128])):
        sa = pm.PointnetSAModule(mlp=list(mlp), npoint=1024, radius=radius, nsample=ns).to(dev).eval()
        feats = torch.randn(B, 2048, 96, device=dev).transpose(1, 2)
        with torch.no_grad():
            geo = sa.sample_and_query(xyz1)
        cases.append(("SA1 ns%d" % ns, lambda sa=sa, feats=feats, geo=geo: sa(xyz1, feats, geometry=geo),
                      2.0 * (99 * mlp[1] + mlp[1] * mlp[2] + mlp[2] * mlp[3]) * 1024 * ns * B))
    # SA level 2: 1024 -> 512 centres, C = 256, ns 16 / 32
    for ns, radius in ((16, 0.05), (32, 0.1)):
        sa = pm.PointnetSAModule(mlp=[256, 128, 196, 256], npoint=512, radius=radius, nsample=ns).to(dev).eval()
        feats = torch.randn(B, 1024, 256, device=dev).transpose(1, 2)
        with torch.no_grad():
            geo = sa.sample_and_query(xyz)
        cases.append(("SA2 ns%d" % ns, lambda sa=sa, feats=feats, geo=geo: sa(xyz, feats, geometry=geo),
                      2.0 * (259 * 128 + 128 * 196 + 196 * 256) * 512 * ns * B))
    # SA level 3: 512 -> 128 centres, C = 512
    xyz3 = xyz[:, :512].contiguous()
    for ns, radius, mlp in ((16, 0.1, [512, 256, 256, 512]), (32, 0.2, [512, 256, 384, 512])):
        sa = pm.PointnetSAModule(mlp=list(mlp), npoint=128, radius=radius, nsample=ns).to(dev).eval()
        feats = torch.randn(B, 512, 512, device=dev).transpose(1, 2)
        with torch.no_grad():
            geo = sa.sample_and_query(xyz3)
        cases.append(("SA3 ns%d" % ns, lambda sa=sa, feats=feats, geo=geo: sa(xyz3, feats, geometry=geo),
                      2.0 * (515 * mlp[1] + mlp[1] * mlp[2] + mlp[2] * mlp[3]) * 128 * ns * B))
    # FP level 0: 12288 <- 2048, C2 = 256, C1 = 6; FP level 1: 2048 <- 1024, 512 + 96
    # FP level 2: 1024 <- 512, 512 + 256; FP level 3: 512 <- 128, 1024 + 512 (layer-by-layer split GEMM, csrc/split_gemm.hip)
    for name, n, m, c2, c1, mlp in (("FP0", 12288, 2048, 256, 6, [262, 128, 128]), ("FP1", 2048, 1024, 512, 96, [608, 256, 256]),
                                    ("FP2", 1024, 512, 512, 256, [768, 512, 512]), ("FP3", 512, 128, 1024, 512, [1536, 512, 512])):
        fp = pm.PointnetFPModule(mlp=mlp).to(dev).eval()
        fp._point_major_out = True
        unk = torch.from_numpy(np.stack([synth.synth_frame(frame=i, n_pts=n, n_obj=256)["pcld"] for i in range(B)])).to(dev)
        kn = unk[:, :m].contiguous()
        kf = torch.randn(B, m, c2, device=dev).transpose(1, 2)
        uf = torch.randn(B, n, c1 + 3, device=dev)[:, :, 3:].transpose(1, 2) if c1 < 32 else torch.randn(B, n, c1, device=dev).transpose(1, 2)
        with torch.no_grad():
            nb = fp.neighbours(unk, kn)
            if name in ("FP2", "FP3"):
                _fused_mlp.MLP_ARITH = "fp32"
                want = fp(unk, kn, uf, kf, neighbours=nb)
                _fused_mlp.MLP_ARITH = "fp16x2"
                got = fp(unk, kn, uf, kf, neighbours=nb)
                print(name, "split GEMM vs fp32 chain: max |diff| / scale = %.2e" % (
                    float((got - want).abs().max()) / max(1.0, float(want.abs().max()))))
        cases.append((name, lambda fp=fp, unk=unk, kn=kn, uf=uf, kf=kf, nb=nb: fp(unk, kn, uf, kf, neighbours=nb),
                      2.0 * sum(a * b for a, b in zip(mlp[:-1], mlp[1:])) * n * B))
    for name, fn, flops in cases:
        res = {}
        outs = {}
        for arith in ("fp32", "bf16x3", "fp16x2"):
            _fused_mlp.MLP_ARITH = arith
            with torch.no_grad():
                res[arith] = ms_of(fn)
                o = fn()
                outs[arith] = (o[1] if isinstance(o, tuple) else o).clone()
        extra = ""
        if name.startswith(("SA0", "SA1")):
            from pvn3d_amd.lib.pointnet2_utils import _ext
            _ext.NARROW_KERNELS = False
            with torch.no_grad():
                extra = "   [narrow-chain kernel off: %7.3f ms]" % ms_of(fn)
            _ext.NARROW_KERNELS = True
        sc = max(1.0, float(outs["fp32"].abs().max()))
        print("%-9s fp32 mfma %7.3f ms (%6.1f TF/s)   bf16x3 %7.3f ms (%6.1f)   fp16x2 %7.3f ms (%6.1f TF/s fp32-equivalent)   "
              "max|diff|/scale vs fp32 chain: bf16x3 %.1e fp16x2 %.1e   dbg=%s" % (
                  name, res["fp32"], flops / res["fp32"] / 1e9, res["bf16x3"], flops / res["bf16x3"] / 1e9,
                  res["fp16x2"], flops / res["fp16x2"] / 1e9, float((outs["bf16x3"] - outs["fp32"]).abs().max()) / sc,
                  float((outs["fp16x2"] - outs["fp32"]).abs().max()) / sc, os.environ.get("PVN3D_S3_DBG", "0")) + extra)
    _fused_mlp.MLP_ARITH = "fp16x2"


if __name__ == "__main__":
    main()
