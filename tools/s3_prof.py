#!/usr/bin/env python3
"""Phase timeline of workgroup 0 of the split-bf16 SA level-2 kernel (PVN3D_S3_DBG & 64 stamps, csrc/sa_mlp_split.hip).
Needs the tuning build: tools/build_probe_lib.sh, then PVN3D_HIP_LIB=tools/libpvn3d_probe.so python tools/s3_prof.py"""
import ctypes
import os
import sys

os.environ["PVN3D_S3_DBG"] = str(64 | int(os.environ.get("S3_EXTRA", "0")))
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd.lib.pointnet2_utils import pointnet2_modules as pm  # noqa: E402
from pvn3d_amd import synth  # noqa: E402
from pvn3d_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
B = 64
xyz = torch.from_numpy(np.stack([synth.synth_frame(frame=i, n_pts=1024, n_obj=256)["pcld"] for i in range(B)])).to(dev)
case = sys.argv[1] if len(sys.argv) > 1 else "sa2"
if case in ("sa1", "sa2"):
    if case == "sa1":
        xyz = torch.from_numpy(np.stack([synth.synth_frame(frame=i, n_pts=2048, n_obj=512)["pcld"] for i in range(B)])).to(dev)
        sa = pm.PointnetSAModule(mlp=[96, 64, 96, 128], npoint=512, radius=0.1, nsample=32).to(dev).eval()
        feats = torch.randn(B, 2048, 96, device=dev).transpose(1, 2)
    else:
        sa = pm.PointnetSAModule(mlp=[256, 128, 196, 256], npoint=512, radius=0.1, nsample=32).to(dev).eval()
        feats = torch.randn(B, 1024, 256, device=dev).transpose(1, 2)
    with torch.no_grad():
        geo = sa.sample_and_query(xyz)
        for _ in range(3):
            sa(xyz, feats, geometry=geo)
else:
    n, m, c2, c1, mlp = (12288, 2048, 256, 6, [262, 128, 128]) if case == "fp0" else (2048, 1024, 512, 96, [608, 256, 256])
    fp = pm.PointnetFPModule(mlp=mlp).to(dev).eval()
    fp._point_major_out = case != "fp0"
    unk = torch.from_numpy(np.stack([synth.synth_frame(frame=i, n_pts=n, n_obj=256)["pcld"] for i in range(B)])).to(dev)
    kn = unk[:, :m].contiguous()
    kf = torch.randn(B, m, c2, device=dev).transpose(1, 2)
    uf = torch.randn(B, n, c1 + 3, device=dev)[:, :, 3:].transpose(1, 2) if c1 < 32 else torch.randn(B, n, c1, device=dev).transpose(1, 2)
    with torch.no_grad():
        nb = fp.neighbours(unk, kn)
        for _ in range(3):
            fp(unk, kn, uf, kf, neighbours=nb)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
f = lib.pvn3d_debug_s3_prof_read          # exported by the -DPVN3D_S3_TUNING build only
f.argtypes = [ctypes.c_void_p]
assert f(buf) == 0
t = np.array(buf[:], dtype=np.int64)
names = ["start", "layer0 done", "bar", "store0+bar", "layer1 done", "bar+store1+bar", "(layer 0: identity chunks done)", "epilogue done"]
t0 = t[0]
for blk in range(4):
    row = t[blk * 8:blk * 8 + 8] - t0
    print("block %d:" % blk, "  ".join("%s %d" % (n, v) for n, v in zip(names, row)))
    print("        deltas:", np.diff(row))
row = t[32:40] - t[32]
print("layer 0 of block 40, MFMA wave 0, cycles since phase I started: identity chunk k done %s, layer 0 done %d" % (
    [int(v) for v in row[1:7]], int(row[7])))
rt = t[128:256]
dc, dr = float(t[3 * 8 + 7] - t[0]), float(rt[3 * 8 + 7] - rt[0])
print("four blocks: %.0f cycles in %.2f us on the real-time counter -> shader clock %.2f GHz" % (dc, dr / 100.0, dc / max(dr, 1.0) / 10.0))
print("loader wave 0 (wait-start, wait-end, chunk-done) x chunks, relative:")
l = t[64:64 + 30].reshape(-1, 3) - t0
print(l)
