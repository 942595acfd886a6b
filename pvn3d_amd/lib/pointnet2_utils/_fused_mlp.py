"""Host side of the fused set-abstraction / feature-propagation kernels (csrc/sa_mlp.hip):
BatchNorm folding and weight packing for a SharedMLP
(pvn3d/lib/utils/etw_pytorch_utils/pytorch_utils.py:25-50), cached per module.

A layer is eligible when it is exactly [1x1 Conv2d] -> [BatchNorm2d in eval mode]? -> ReLU
(the only form PVN3D's Pointnet2MSG builds).  Anything else makes ``pack_shared_mlp`` return
None and the caller keeps the unfused torch path.
"""
import ctypes
import math
import os

import torch
import torch.nn as nn


# Bumped by invalidate_packed(): part of every cache signature below.  The per-tensor signature
# (data_ptr, _version) does not see writes that bypass autograd's version counter (collectives, `.data`
# updates), so code that changes weights that way calls invalidate_packed() -- sharding.broadcast_parameters
# and train_step do.
_WEIGHTS_EPOCH = [0]


def invalidate_packed():
    """Drop every cached folded/packed SharedMLP: call after changing weights or BatchNorm buffers through
    a path that does not bump tensor._version (torch.distributed collectives, `.data` writes)."""
    _WEIGHTS_EPOCH[0] += 1


# Arithmetic of the fused inference chains where a chain has both kernels (csrc/sa_mlp_split.hip decides per shape):
#   "bf16x3": every fp32 operand as the exact sum of three bf16 pieces, six partial products per multiply accumulated in
#             fp32 on the bf16 matrix pipe -- fp32 accuracy (the dropped terms are below fp32's own rounding step) at
#             6/16 of the fp32-MFMA cost;
#   "fp32":   v_mfma_f32_32x32x2_f32 everywhere (csrc/sa_mlp.hip).
#   "fp16x2": (round 5, default) two fp16 pieces per operand, three partial products on the fp16 matrix pipe, operands
#             scaled by exact powers of two into fp16's range -- as close to fp64 as the fp32 FMA chain (measured), half the
#             matrix-pipe time of "bf16x3"; chains the fp16 x 2 kernels do not take fall back to "bf16x3" behaviour.
MLP_ARITH = os.environ.get("PVN3D_MLP_ARITH", "fp16x2")


def split_arith():
    """True when a split (bf16 x 3 or fp16 x 2) arithmetic is selected, i.e. everything but "fp32"."""
    return MLP_ARITH in ("bf16x3", "fp16x2")


class PackedMLP(object):
    """Device buffers + the host pointer arrays pvn3d_sa_mlp_maxpool / pvn3d_fp_interp_mlp take."""

    def __init__(self, dims, w_list, b_list, folded=None):
        self.dims = list(dims)
        self.n_layers = len(w_list)
        self.w = w_list            # keep the tensors alive
        self.b = b_list
        self.dims_c = (ctypes.c_int * len(dims))(*dims)
        self.w_c = (ctypes.c_void_p * self.n_layers)(*[t.data_ptr() for t in w_list])
        self.b_c = (ctypes.c_void_p * self.n_layers)(*[t.data_ptr() for t in b_list])
        self._folded = folded      # [(W' (M, K) fp32 in the kernels' layer-0 channel order)]: source of the split packing
        self._split = None

    def precontracted(self, c_feat):
        """The chain with the GATHERED half of its first conv taken out (the caller applies it per source point, before
        the gather / interpolation: _ext.sa_precontract, _ext.fp_interp_mlp): layer 0 becomes [I | Wr] over
        [Wf.f (M0 channels); the rest (SA: relative xyz, FP: skip features)], the other layers are unchanged.
        -> (PackedMLP for dims [M0 + rest, M0, ...], Wf (M0, c_feat) fp32)."""
        cache = getattr(self, "_pre", None)
        if cache is not None and cache[0] == c_feat:
            return cache[1], cache[2]
        W0 = self._folded[0]                                  # kernel order: features first, then the xyz columns
        m0 = W0.shape[0]
        assert W0.shape[1] > c_feat
        w0 = torch.cat([torch.eye(m0, dtype=torch.float32, device=W0.device), W0[:, c_feat:]], 1).contiguous()
        folded = [w0] + list(self._folded[1:])
        pre = PackedMLP([w0.shape[1]] + self.dims[1:], [_pack_weight(W) for W in folded], self.b, folded=folded)
        self._pre = (c_feat, pre, W0[:, :c_feat].contiguous())
        return pre, self._pre[2]

    def s16(self, c_first):
        """Two-layer chain for csrc/split_gemm.hip, built on first use: layer 0 split at input channel `c_first`
        (W = [Wa | Wb]: interpolated channels | skip channels), every matrix in the s16 layout.
        -> dict(wa, wb, w2 int16 buffers, b1, b2 fp32 biases padded to 128, n1, n2, s_a, s_b, s_h slab counts)."""
        cache = getattr(self, "_s16", None)
        if cache is not None and cache[0] == c_first:
            return cache[1]
        assert self.n_layers == 2
        (W1, W2) = self._folded
        n1, n2 = W1.shape[0], W2.shape[0]
        s_a, s_b, s_h = _slabs(c_first), _slabs(W1.shape[1] - c_first), _slabs(n1)

        def pad_bias(b, n):
            out = torch.zeros(((n + 127) // 128) * 128, dtype=torch.float32, device=b.device)
            out[:n] = b[:n]
            return out
        d = dict(wa=_pack_weight_s16(W1[:, :c_first], s_a), wb=_pack_weight_s16(W1[:, c_first:], s_b),
                 w2=_pack_weight_s16(W2, s_h), b1=pad_bias(self.b[0], n1), b2=pad_bias(self.b[1], n2),
                 n1=n1, n2=n2, s_a=s_a, s_b=s_b, s_h=s_h)
        self._s16 = (c_first, d)
        return d

    def h16(self, c_first):
        """The two-layer chain for csrc/split_gemm.hip in the fp16 x 2 arithmetic (pvn3d_split_gemm2), built on first
        use: layer 0 split at input channel `c_first` (W = [Wa | Wb]), every matrix in the h16 layout with its own
        power-of-two scale.  -> dict(wa, wb, w2 int16 buffers; sw_a, sw_b, sw_2 scales; na, nb = ||Wa||_inf, ||Wb||_inf,
        b1max; b1, b2 fp32 biases padded to 128; n1, n2; s_a, s_b, s_h slab counts)."""
        cache = getattr(self, "_h16", None)
        if cache is not None and cache[0] == c_first:
            return cache[1]
        assert self.n_layers == 2
        (W1, W2) = self._folded
        n1, n2 = W1.shape[0], W2.shape[0]
        s_a, s_b, s_h = _slabs(c_first), _slabs(W1.shape[1] - c_first), _slabs(n1)

        def pad_bias(b, n):
            out = torch.zeros(((n + 127) // 128) * 128, dtype=torch.float32, device=b.device)
            out[:n] = b[:n]
            return out
        Wa, Wb = W1[:, :c_first], W1[:, c_first:]
        sw_a, sw_b, sw_2 = _pow2_weight_scale(Wa), _pow2_weight_scale(Wb), _pow2_weight_scale(W2)
        d = dict(wa=_pack_weight_h16(Wa * sw_a, s_a), wb=_pack_weight_h16(Wb * sw_b, s_b), w2=_pack_weight_h16(W2 * sw_2, s_h),
                 sw_a=sw_a, sw_b=sw_b, sw_2=sw_2, na=float(Wa.abs().sum(1).max()), nb=float(Wb.abs().sum(1).max()),
                 b1max=float(self.b[0].abs().max()), b1=pad_bias(self.b[0], n1), b2=pad_bias(self.b[1], n2),
                 n1=n1, n2=n2, s_a=s_a, s_b=s_b, s_h=s_h)
        self._h16 = (c_first, d)
        return d

    def split2(self):
        """-> (ctypes array of the fp16 x 2 weight buffers, ctypes float[3 * n_layers] layer_meta) for
        pvn3d_*_split2 (csrc/sa_mlp_split.hip, AR = 1), built on first use: per layer a power-of-two weight scale sw
        with max|sw W'| in [2^13, 2^14], the two fp16 pieces of sw W' (round to nearest), ||W'||_inf and max|bias|."""
        if getattr(self, "_split2", None) is None:
            ws, meta = [], []
            for W, b in zip(self._folded, self.b):
                sw = _pow2_weight_scale(W)
                ws.append(_pack_weight_split2(W * sw))
                meta += [sw, float(W.abs().sum(1).max()), float(b.abs().max())]
            self._split2 = (ws, (ctypes.c_void_p * self.n_layers)(*[t.data_ptr() for t in ws]),
                            (ctypes.c_float * len(meta))(*meta))
        return self._split2[1], self._split2[2]

    def split(self):
        """-> ctypes array of the split-bf16 weight buffers (csrc/sa_mlp_split.hip), built on first use."""
        if self._split is None:
            ws = [_pack_weight_split(W) for W in self._folded]
            self._split = (ws, (ctypes.c_void_p * self.n_layers)(*[t.data_ptr() for t in ws]))
        return self._split[1]


def _pack_weight_split(W):
    """W (M, K) float32 -> int16 [ceil(K/16) slabs][ceil(M/32) row tiles][3 pieces][64 lanes][8]: the three bf16 pieces
    of W (each rounded to nearest: hi = bf16(W), mid = bf16(W - hi), lo = bf16(W - hi - mid); W - hi - mid - lo is at
    most 2^-26 |W|), lane l of a fragment holding row mt*32 + (l & 31), k = 16*slab + 8*(l >> 5) + 0..7; zero outside
    M x K (include/pvn3d_hip.h)."""
    M, K = W.shape
    MT, S = (M + 31) // 32, (K + 15) // 16
    Wp = torch.zeros((MT * 32, S * 16), dtype=torch.float32, device=W.device)
    Wp[:M, :K] = W
    hi = Wp.to(torch.bfloat16)
    r1 = Wp - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    pieces = torch.stack([hi, mid, lo], 0)                      # (3, MT*32, S*16)
    # (piece, mt, r, s, half, j) -> (s, mt, piece, half, r, j)
    out = pieces.view(3, MT, 32, S, 2, 8).permute(3, 1, 0, 4, 2, 5).contiguous()
    return out.view(torch.int16).view(S, MT, 3, 64, 8)


def _pow2_weight_scale(W):
    """The power of two that puts max|W| into (2^13, 2^14] (1 for a zero matrix)."""
    wmax = float(W.abs().max()) if W.numel() else 0.0
    if not wmax > 0:
        return 1.0
    sw = 2.0 ** (14 - math.ceil(math.log2(wmax)))
    while wmax * sw > 16384.0:
        sw *= 0.5
    return sw


def _pack_weight_h16(W, slabs):
    """W (M, K) float32, already scaled into fp16's range -> int16 [roundup128(M)][slabs][2 pieces][16]: the h16 layout of
    include/pvn3d_hip.h (hi = fp16(W), lo = fp16(W - hi), round to nearest), zero outside M x K."""
    M, K = W.shape
    Mp = ((M + 127) // 128) * 128
    assert K <= 16 * slabs
    Wp = torch.zeros((Mp, slabs * 16), dtype=torch.float32, device=W.device)
    Wp[:M, :K] = W
    hi = Wp.to(torch.float16)
    lo = (Wp - hi.float()).to(torch.float16)
    pieces = torch.stack([hi, lo], 0)                            # (2, Mp, slabs*16)
    out = pieces.view(2, Mp, slabs, 16).permute(1, 2, 0, 3).contiguous()
    return out.view(torch.int16)


def _pack_weight_split2(W):
    """W (M, K) float32, already scaled into fp16's range -> int16 [ceil(K/16)][ceil(M/32)][2 pieces][64 lanes][8]:
    hi = fp16(W), lo = fp16(W - hi), both rounded to nearest; fragment order as _pack_weight_split."""
    M, K = W.shape
    MT, S = (M + 31) // 32, (K + 15) // 16
    Wp = torch.zeros((MT * 32, S * 16), dtype=torch.float32, device=W.device)
    Wp[:M, :K] = W
    hi = Wp.to(torch.float16)
    lo = (Wp - hi.float()).to(torch.float16)
    pieces = torch.stack([hi, lo], 0)                           # (2, MT*32, S*16)
    out = pieces.view(2, MT, 32, S, 2, 8).permute(3, 1, 0, 4, 2, 5).contiguous()
    return out.view(torch.int16).view(S, MT, 2, 64, 8)


def _slabs(k):
    """16-k slabs of a contraction of k channels, rounded up to whole 32-k chunks"""
    return ((k + 31) // 32) * 2


def _pack_weight_s16(W, slabs):
    """W (M, K) float32 -> int16 [roundup128(M)][slabs][3 pieces][16]: the s16 layout of include/pvn3d_hip.h, pieces
    rounded to nearest (hi = bf16(W), mid = bf16(W - hi), lo = bf16(W - hi - mid)), zero outside M x K."""
    M, K = W.shape
    Mp = ((M + 127) // 128) * 128
    assert K <= 16 * slabs
    Wp = torch.zeros((Mp, slabs * 16), dtype=torch.float32, device=W.device)
    Wp[:M, :K] = W
    hi = Wp.to(torch.bfloat16)
    r1 = Wp - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    pieces = torch.stack([hi, mid, lo], 0)                       # (3, Mp, slabs*16)
    out = pieces.view(3, Mp, slabs, 16).permute(1, 2, 0, 3).contiguous()
    return out.view(torch.int16)


def _pack_weight(W):
    """W (M,K) float32 -> [ceil(K/4)][ceil(M/32)][64][2]: entry (k4, mt, lane, j) =
    W[mt*32 + (lane & 31)][4*k4 + 2*j + (lane >> 5)], zero outside (include/pvn3d_hip.h)."""
    M, K = W.shape
    MT, K4 = (M + 31) // 32, (K + 3) // 4
    Wp = torch.zeros((MT * 32, K4 * 4), dtype=torch.float32, device=W.device)
    Wp[:M, :K] = W
    # (mt, r, k4, j, half) -> (k4, mt, half, r, j)
    return Wp.view(MT, 32, K4, 2, 2).permute(2, 0, 4, 1, 3).contiguous().view(K4, MT, 64, 2)


def _fold(layer):
    """One SharedMLP layer (nn.Sequential of conv / normlayer / activation) -> (W', b') or None."""
    conv = getattr(layer, "conv", None)
    if not isinstance(conv, nn.Conv2d) or conv.kernel_size != (1, 1) or conv.stride != (1, 1) \
            or conv.padding != (0, 0) or conv.groups != 1:
        return None
    names = [n for n, _ in layer.named_children()]
    if names and names[0] != "conv":          # pre-activation layout: not handled
        return None
    act = getattr(layer, "activation", None)
    if not isinstance(act, nn.ReLU):
        return None
    W = conv.weight.detach().float().view(conv.out_channels, conv.in_channels)
    b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(conv.out_channels, device=W.device)
    norm = getattr(layer, "normlayer", None)
    if norm is not None:
        bn = getattr(norm, "bn", None)
        if not isinstance(bn, nn.BatchNorm2d) or bn.training or not bn.track_running_stats:
            return None
        g = bn.weight.detach().float() if bn.affine else torch.ones_like(bn.running_var)
        beta = bn.bias.detach().float() if bn.affine else torch.zeros_like(bn.running_var)
        s = g / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        W = W * s[:, None]
        b = (b - bn.running_mean.detach().float()) * s + beta
    return W, b


def pack_shared_mlp(mlp, max_width=512, n_xyz_first=0):
    """SharedMLP -> PackedMLP (cached on the module until a parameter/buffer changes), or None.
    n_xyz_first: the first n_xyz_first input channels of layer 0 (the reference puts the relative
    xyz in front of the features, pointnet2_utils.py:319-321) are moved behind the features,
    the channel order pvn3d_sa_mlp_maxpool gathers in."""
    sig = []
    for t in list(mlp.parameters()) + list(mlp.buffers()):
        sig.append((t.data_ptr(), t._version))
    sig = (tuple(sig), mlp.training, n_xyz_first, _WEIGHTS_EPOCH[0])
    cache = getattr(mlp, "_pvn3d_packed", None)
    if cache is not None and cache[0] == sig:
        return cache[1]
    folded = []
    for layer in mlp.children():
        f = _fold(layer)
        if f is None:
            mlp._pvn3d_packed = (sig, None)
            return None
        folded.append(f)
    if not folded or len(folded) > 4:
        mlp._pvn3d_packed = (sig, None)
        return None
    dims = [folded[0][0].shape[1]] + [W.shape[0] for W, _ in folded]
    if max(dims[1:]) > max_width or any(folded[i][0].shape[1] != dims[i] for i in range(len(folded))):
        mlp._pvn3d_packed = (sig, None)
        return None
    w_list, b_list, w_folded = [], [], []
    for li, (W, b) in enumerate(folded):
        if li == 0 and n_xyz_first:
            W = torch.cat([W[:, n_xyz_first:], W[:, :n_xyz_first]], dim=1)
        w_folded.append(W)
        w_list.append(_pack_weight(W))
        M = W.shape[0]
        bp = torch.zeros(((M + 31) // 32) * 32, dtype=torch.float32, device=W.device)
        bp[:M] = b
        b_list.append(bp)
    packed = PackedMLP(dims, w_list, b_list, folded=w_folded)
    mlp._pvn3d_packed = (sig, packed)
    return packed
