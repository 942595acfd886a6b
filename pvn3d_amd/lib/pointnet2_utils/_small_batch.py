"""Layer-by-layer inference SharedMLP for launches too small for the fused chains (csrc/small_batch.hip).

The fused kernels of csrc/sa_mlp.hip give one workgroup 64 columns and the whole layer chain.  With one frame per
call (the reference's test_mini_batch_size = 1, pvn3d/common.py:41) the deep levels of Pointnet2MSG have 512 - 4096
columns, i.e. 8 - 64 workgroups on 256 CUs.  Below ``MAX_FUSED_WGS`` workgroups the modules run the same layers
(fp32 MFMA, eval BatchNorm folded) one layer per launch with one wave per 32 x 32 output tile instead.
"""
import torch

from ..._lib import lib, check, on_device
from . import _fused_mlp

# a fused launch with fewer 64-column workgroups than this goes layer by layer (0 disables the path)
MAX_FUSED_WGS = 128


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def folded_layers(mlp):
    """SharedMLP -> [(W' (Cout, Cin) fp32 contiguous, b' (Cout))] with eval BatchNorm folded (cached on the
    module like _fused_mlp.pack_shared_mlp), or None when a layer is not conv -> [bn] -> relu."""
    sig = []
    for t in list(mlp.parameters()) + list(mlp.buffers()):
        sig.append((t.data_ptr(), t._version))
    sig = (tuple(sig), mlp.training, _fused_mlp._WEIGHTS_EPOCH[0])
    cache = getattr(mlp, "_pvn3d_folded", None)
    if cache is not None and cache[0] == sig:
        return cache[1]
    out = []
    for layer in mlp.children():
        f = _fused_mlp._fold(layer)
        if f is None:
            out = None
            break
        W, b = f[0].contiguous(), f[1].contiguous()
        if W.size(1) & 3:                         # the layer-0 input rows are zero-padded to a multiple of 4 columns
            Wp = torch.zeros((W.size(0), (W.size(1) + 3) // 4 * 4), dtype=torch.float32, device=W.device)
            Wp[:, :W.size(1)] = W
            W = Wp
        out.append((W, b))
    mlp._pvn3d_folded = (sig, out)
    return out


def _strides3(t):
    return t.data_ptr(), t.stride(0), t.stride(1), t.stride(2)


def _chain(x, layers, st):
    """x (rows, K) fp32 -> relu(W_L ... relu(W_0 x + b_0) ... + b_L) (rows, C_L)."""
    rows = x.size(0)
    for W, b in layers:
        cout, cin = W.shape
        y = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
        splits = int(lib.pvn3d_sb_linear_splits(rows, cout, cin))        # few output tiles: cut K (deterministic)
        part = torch.empty((splits, rows, cout), dtype=torch.float32, device=x.device) if splits > 1 else None
        check(lib.pvn3d_sb_linear(rows, cout, cin, x.data_ptr(), x.size(1), W.data_ptr(), cin, b.data_ptr(), 1,
                                  y.data_ptr(), cout, part.data_ptr() if part is not None else None, splits, st),
              "sb_linear")
        x = y
    return x


def sa_scale(xyz, new_xyz, features, idx, use_xyz, layers, out_pm, out_coff):
    """One SA scale: gather -> layers -> max over nsample, written into out_pm[:, :, out_coff : out_coff + C_L]."""
    B, N, m, ns = xyz.size(0), xyz.size(1), idx.size(1), idx.size(2)
    C = features.size(1) if features is not None else 0
    c0 = (3 if use_xyz else 0) + C
    ld0 = (c0 + 3) // 4 * 4
    dev = xyz.device
    with on_device(dev):
        st = _stream(xyz)
        x0 = torch.empty((B * m * ns, ld0), dtype=torch.float32, device=dev)
        fp, fsb, fsc, fsn = _strides3(features) if features is not None else (None, 0, 0, 0)
        check(lib.pvn3d_sb_gather_sa(B, N, m, ns, C, 1 if use_xyz else 0, xyz.data_ptr(), new_xyz.data_ptr(), fp, fsb, fsc,
                                     fsn, idx.data_ptr(), x0.data_ptr(), ld0, st), "sb_gather_sa")
        assert layers[0][0].size(1) == ld0
        h = _chain(x0, layers, st)
        cl = h.size(1)
        check(lib.pvn3d_sb_pool_max(B * m, ns, cl, cl, h.data_ptr(), out_pm.data_ptr() + 4 * out_coff, out_pm.size(2), st),
              "sb_pool_max")


def fp_module(known_feats, unknow_feats, idx, weight, layers, point_major_out):
    """interpolate ++ skip features -> layers.  Returns (B, C_L, n): a transposed view of the point-major result
    (point_major_out) or contiguous."""
    B, C2, mk = known_feats.shape
    n = idx.size(1)
    C1 = unknow_feats.size(1) if unknow_feats is not None else 0
    c0 = C2 + C1
    ld0 = (c0 + 3) // 4 * 4
    dev = known_feats.device
    with on_device(dev):
        st = _stream(known_feats)
        x0 = torch.empty((B * n, ld0), dtype=torch.float32, device=dev)
        kp, ksb, ksc, ksn = _strides3(known_feats)
        up, usb, usc, usn = _strides3(unknow_feats) if unknow_feats is not None else (None, 0, 0, 0)
        check(lib.pvn3d_sb_gather_fp(B, n, mk, C2, C1, kp, ksb, ksc, ksn, up, usb, usc, usn, idx.data_ptr(),
                                     weight.data_ptr(), x0.data_ptr(), ld0, st), "sb_gather_fp")
        assert layers[0][0].size(1) == ld0
        h = _chain(x0, layers, st)
    out = h.view(B, n, h.size(1)).transpose(1, 2)
    return out if point_major_out else out.contiguous()
