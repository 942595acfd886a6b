import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from pvn3d_amd._lib import lib, check
from pvn3d_amd.lib.pointnet2_utils import _fused_mlp as fm, pointnet2_modules as pm, _ext
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, n, m, c2, c1, mlp = 64, 1024, 512, 512, 256, [768, 512, 512]
fp = pm.PointnetFPModule(mlp=mlp).to(dev).eval()
fp._point_major_out = True
unk = torch.rand(B, n, 3, device=dev); kn = unk[:, :m].contiguous()
kf = torch.randn(B, m, c2, device=dev).transpose(1, 2)
uf = torch.randn(B, n, c1, device=dev).transpose(1, 2)
with torch.no_grad():
    idx, wgt = fp.neighbours(unk, kn)
    packed = fm.pack_shared_mlp(fp.mlp)
    W1, W2 = [w.double() for w in packed._folded]
    b1, b2 = packed.b[0][:512].double(), packed.b[1][:512].double()
    bi = torch.arange(B, device=dev)[:, None, None]
    kfp = kf.transpose(1, 2).double()            # (B, m, c2)
    interp = (kfp[bi, idx.long()] * wgt.double()[..., None]).sum(2)   # (B, n, c2)
    x = torch.cat([interp, uf.transpose(1, 2).double()], 2)
    h = torch.relu(x @ W1.T + b1)
    want = torch.relu(h @ W2.T + b2)            # (B, n, 512)
    fm.MLP_ARITH = "fp32"
    a = fp(unk, kn, uf, kf, neighbours=(idx, wgt)).transpose(1, 2).double()
    fm.MLP_ARITH = "bf16x3"
    g = fp(unk, kn, uf, kf, neighbours=(idx, wgt)).transpose(1, 2).double()
    print("fp32 chain err", float((a - want).abs().max()), "layerwise err", float((g - want).abs().max()), "scale", float(want.abs().max()))
    e = (g - want).abs()
    print("err by frame", e.amax((1, 2)).cpu().numpy())
    print("err by point block of 128 (frame 0)", e[0].amax(1).view(-1, 128).amax(1).cpu().numpy())
with torch.no_grad():
    g2 = fp(unk, kn, uf, kf, neighbours=(idx, wgt)).transpose(1, 2).double()
    print("run-to-run identical:", bool((g2 == g).all()), "second run err", float((g2 - want).abs().max()))
    # stage by stage
    w = packed.s16(c2)
    st = torch.cuda.current_stream().cuda_stream
    P, Pk = B * n, B * m
    kfb, ufb = kf.transpose(1, 2).contiguous(), uf.transpose(1, 2).contiguous()
    xk = torch.empty(Pk * w["s_a"] * 96, dtype=torch.uint8, device=dev)
    xu = torch.empty(P * w["s_b"] * 96, dtype=torch.uint8, device=dev)
    z = torch.empty(Pk, 512, device=dev)
    hb = torch.empty(P * w["s_h"] * 96, dtype=torch.uint8, device=dev)
    out = torch.empty(P, 512, device=dev)
    check(lib.pvn3d_split_rows(Pk, c2, kfb.data_ptr(), c2, xk.data_ptr(), w["s_a"], st), "a")
    check(lib.pvn3d_split_rows(P, c1, ufb.data_ptr(), c1, xu.data_ptr(), w["s_b"], st), "b")
    check(lib.pvn3d_split_gemm(Pk, 512, w["s_a"], xk.data_ptr(), w["wa"].data_ptr(), None, 0, None, 0, 0, 0, None, None, z.data_ptr(), 512, None, 0, st), "z")
    zw = kfp.reshape(Pk, c2) @ W1[:, :c2].T
    print("Z err", float((z.double() - zw).abs().max()))
    check(lib.pvn3d_split_gemm(P, 512, w["s_b"], xu.data_ptr(), w["wb"].data_ptr(), w["b1"].data_ptr(), 1, z.data_ptr(), 512, n, m, idx.data_ptr(), wgt.data_ptr(), out.data_ptr(), 512, hb.data_ptr(), w["s_h"], st), "h")
    print("H err (fp32 out)", float((out.double() - h.reshape(P, 512)).abs().max()))
    v = (hb.view(torch.int16).view(P, w["s_h"], 3, 16).to(torch.int32) << 16).view(torch.float32).double().sum(2).reshape(P, 512)
    print("H err (s16 out)", float((v - h.reshape(P, 512)).abs().max()))
    check(lib.pvn3d_split_gemm(P, 512, w["s_h"], hb.data_ptr(), w["w2"].data_ptr(), w["b2"].data_ptr(), 1, None, 0, 0, 0, None, None, out.data_ptr(), 512, None, 0, st), "o")
    print("out err", float((out.double() - want.reshape(P, 512)).abs().max()))
with torch.no_grad():
    ref_plain = torch.relu(ufb.reshape(P, c1).double() @ W1[:, c2:].T + b1)
    for trial in range(3):
        out.zero_()
        check(lib.pvn3d_split_gemm(P, 512, w["s_b"], xu.data_ptr(), w["wb"].data_ptr(), w["b1"].data_ptr(), 1, None, 0, 0, 0, None, None, out.data_ptr(), 512, None, 0, st), "h")
        e = (out.double() - ref_plain).abs().amax(1).view(-1, 128).amax(1)
        print("plain GEMM P=65536 K=256: err", float(e.max()), "bad tiles", int((e > 1e-3).sum()), (e > 1e-3).nonzero().flatten()[:10].cpu().numpy())
    zz = torch.zeros_like(z)
    for trial in range(2):
        check(lib.pvn3d_split_gemm(P, 512, w["s_b"], xu.data_ptr(), w["wb"].data_ptr(), w["b1"].data_ptr(), 1, zz.data_ptr(), 512, n, m, idx.data_ptr(), wgt.data_ptr(), out.data_ptr(), 512, None, 0, st), "h")
        e = (out.double() - ref_plain).abs().amax(1).view(-1, 128).amax(1)
        print("gather of zeros: err", float(e.max()), "bad tiles", int((e > 1e-3).sum()))
with torch.no_grad():
    # which Z elements does the H kernel see wrong?  Feed identity-like setup: X = 0, bias = 0, relu off: H = interp(Z)
    xz = torch.zeros_like(xu)
    for trial in range(3):
        out.fill_(float("nan"))
        check(lib.pvn3d_split_gemm(P, 512, w["s_b"], xz.data_ptr(), w["wb"].data_ptr(), None, 0, z.data_ptr(), 512, n, m, idx.data_ptr(), wgt.data_ptr(), out.data_ptr(), 512, None, 0, st), "h")
        print("NaN left in out:", int(torch.isnan(out).sum()))
        f = (torch.arange(P, device=dev) // n).long()
        ref = sum(z.double()[f * m + idx.reshape(P, 3)[:, t].long()] * wgt.reshape(P, 3)[:, t:t + 1].double() for t in range(3))
        bad = ~((out.double() - ref).abs() <= 1e-4)
        if bad.any():
            pts0 = bad.any(1).nonzero().flatten()[0]
            ch0 = bad[pts0].nonzero().flatten()[0]
            print("  example: got", float(out[pts0, ch0]), "want", float(ref[pts0, ch0]), "z rows", idx.reshape(P, 3)[pts0].cpu().numpy(), "w", wgt.reshape(P, 3)[pts0].cpu().numpy(),
                  "z vals", [float(z[(pts0 // n) * m + idx.reshape(P, 3)[pts0, t].long(), ch0]) for t in range(3)])
        print("interp(Z) only: bad elements", int(bad.sum()), "of", bad.numel(), "bad points", int(bad.any(1).sum()))
        if bad.any():
            pts = bad.any(1).nonzero().flatten()
            print("  first bad points", pts[:12].cpu().numpy(), "channels bad in first:", bad[pts[0]].nonzero().flatten()[[0, -1]].cpu().numpy(), int(bad[pts[0]].sum()))
            print("  bad point index mod 128 histogram (16 bins)", torch.histc((pts % 128).float(), bins=16, min=0, max=128).cpu().numpy())
