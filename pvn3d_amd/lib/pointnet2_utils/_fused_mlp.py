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

    def __init__(self, dims, w_list, b_list, folded=None, equil=None):
        # equil: (rinv,) when `folded` / `b_list` already are a diagonally rescaled chain (equilibrate() below) whose
        # last layer's rows are undone by multiplying the outputs with rinv -- what precontracted(..., equil=True) builds
        self._equil_given = equil
        self.dims = list(dims)
        self.n_layers = len(w_list)
        self.w = w_list            # keep the tensors alive
        self.b = b_list
        self.dims_c = (ctypes.c_int * len(dims))(*dims)
        self.w_c = (ctypes.c_void_p * self.n_layers)(*[t.data_ptr() for t in w_list])
        self.b_c = (ctypes.c_void_p * self.n_layers)(*[t.data_ptr() for t in b_list])
        self._folded = folded      # [(W' (M, K) fp32 in the kernels' layer-0 channel order)]: source of the split packing
        self._split = None

    def precontracted(self, c_feat, equil=False):
        """The chain with the GATHERED half of its first conv taken out (the caller applies it per source point, before
        the gather / interpolation: _ext.sa_precontract, _ext.fp_interp_mlp): layer 0 becomes [I | Wr] over
        [Wf.f (M0 channels); the rest (SA: relative xyz, FP: skip features)], the other layers are unchanged.
        -> (PackedMLP for dims [M0 + rest, M0, ...], Wf (M0, c_feat) fp32).
        equil (the fp16 x 2 arithmetic): both halves come from the diagonally rescaled chain (equilibrated()): Wf = D_0 Wf,
        layer 0 = [I | D_0 Wr] with bias D_0 b_0 -- the identity block stays an identity because the pre-contraction GEMM
        delivers D_0 Wf.f -- and the returned chain carries the output multipliers D_L^-1."""
        cache = getattr(self, "_pre", None)
        if cache is not None and cache[0] == (c_feat, bool(equil)):
            return cache[1], cache[2]
        if equil:
            Ws, bs, rinv = self.equilibrated()
            b_list = [_pad32(b) for b in bs]
        else:
            Ws, b_list, rinv = self._folded, self.b, None
        W0 = Ws[0]                                            # kernel order: features first, then the xyz columns
        m0 = W0.shape[0]
        assert W0.shape[1] > c_feat
        w0 = torch.cat([torch.eye(m0, dtype=torch.float32, device=W0.device), W0[:, c_feat:]], 1).contiguous()
        folded = [w0] + list(Ws[1:])
        pre = PackedMLP([w0.shape[1]] + self.dims[1:], [_pack_weight(W) for W in folded], b_list, folded=folded,
                        equil=(rinv,) if equil else None)
        pre.identity_a = True          # layer 0 = [I | Wr]: the low fp16 pieces of its table-A slabs are zero (PVN3D_MLP_IDENTITY_A)
        self._pre = ((c_feat, bool(equil)), pre, W0[:, :c_feat].contiguous())
        return pre, self._pre[2]

    def fp16x2_safe(self):
        """Does the two-piece fp16 arithmetic hold fp32's per-channel accuracy on this chain?  fp16x2_probe() on the rescaled
        chain, cached.  False sends the chain to the bf16 x 3 kernels (_ext): what is left after the diagonal rescaling
        are chains whose weights span more than two fp16 pieces can hold inside one row or column -- e.g. a channel that
        the nominal input never switches on, feeding a huge column of the next layer."""
        if self._equil_given is not None:
            return True          # (a pre-contracted half of a rescaled chain: the verdict was taken on the whole chain)
        if getattr(self, "_safe", None) is None:
            Ws, bs, rinv = self.equilibrated()
            self._safe = fp16x2_verdict(*fp16x2_probe(Ws, bs, rinv), k_first=Ws[0].shape[1])
        return self._safe[0]

    def equilibrated(self):
        """-> ([W~_l], [b~_l], rinv): the chain rescaled diagonally by powers of two (equilibrate() below), cached.  A chain
        built by precontracted(..., equil=True) already is one and returns itself."""
        if self._equil_given is not None:
            return self._folded, [b[:W.shape[0]] for W, b in zip(self._folded, self.b)], self._equil_given[0]
        if getattr(self, "_eq", None) is None:
            self._eq = equilibrate(self._folded, [b[:W.shape[0]] for W, b in zip(self._folded, self.b)])
        return self._eq

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
        use: layer 0 split at input channel `c_first` (W = [Wa | Wb]), every matrix in the h16 layout.  Round 6: every ROW
        of every matrix carries its own power-of-two scale (undone by the GEMM's w_row_mul), and the hidden layer is
        rescaled diagonally (H~ = D_1 H, W2 -> W2 D_1^-1: exact) so that one scalar bound serves all its channels.
        -> dict(wa, wb, w2 int16 buffers; rm_a, rm_b, rm_2 device per-row multipliers (padded to 128); na, nb =
        ||D_1 Wa||_inf, ||D_1 Wb||_inf, b1max = max|D_1 b1|; b1 (= D_1 b1), b2 fp32 biases padded to 128; n1, n2; s_a,
        s_b, s_h slab counts)."""
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
        b1, b2 = self.b[0][:n1], self.b[1][:n2]
        lw = W1.abs().double().sum(1) * EQUIL_NOMINAL_INPUT
        ub = torch.clamp(lw + b1.double(), min=0.0)
        live = ub > (lw + b1.abs().double()) * 2.0 ** -10
        r1 = _pow2_inv(torch.maximum(torch.where(live, ub, lw), lw / EQUIL_MAX_BOOST))      # r1 H_o <~ 1 (equilibrate())
        Wa, Wb, b1 = W1[:, :c_first] * r1[:, None], W1[:, c_first:] * r1[:, None], b1 * r1
        W2 = W2 / r1[None, :]
        wa, rm_a = _pack_weight_h16_rows(Wa, s_a)
        wb, rm_b = _pack_weight_h16_rows(Wb, s_b)
        w2, rm_2 = _pack_weight_h16_rows(W2, s_h)
        d = dict(wa=wa, wb=wb, w2=w2, rm_a=rm_a, rm_b=rm_b, rm_2=rm_2,
                 na=float(Wa.abs().sum(1).max()), nb=float(Wb.abs().sum(1).max()),
                 b1max=max(float(b1.max()), 0.0), b1=pad_bias(b1, n1), b2=pad_bias(b2, n2),
                 chain=([torch.cat([Wa, Wb], 1), W2], [b1, b2]),
                 n1=n1, n2=n2, s_a=s_a, s_b=s_b, s_h=s_h)
        self._h16 = (c_first, d)
        return d

    def h16_safe(self, c_first):
        """fp16x2_safe() for the layer-by-layer form (every row with its own weight scale, hidden layer rescaled)."""
        if getattr(self, "_h16_safe", None) is None or self._h16_safe[0] != c_first:
            Ws, bs = self.h16(c_first)["chain"]
            self._h16_safe = (c_first,) + fp16x2_verdict(*fp16x2_probe(Ws, bs, None, row_scaled=True), k_first=Ws[0].shape[1])
        return self._h16_safe[1]

    def split2(self):
        """-> (ctypes array of the fp16 x 2 weight buffers, ctypes float[3 * n_layers] layer_meta, ctypes array of the
        bias buffers, device tensor of the output multipliers) for pvn3d_*_split2 (csrc/sa_mlp_split.hip, AR = 1), built
        on first use from the diagonally rescaled chain (equilibrated(): every row of every layer in fp16's range at full
        two-piece precision, every hidden channel's bound ~1): per layer a power-of-two weight scale sw with max|sw W~| in
        [2^13, 2^14], the two fp16 pieces of sw W~ (round to nearest), ||W~||_inf and max|b~|; the biases b~ padded to 32;
        rinv = D_L^-1 padded to 32, the `out_row_mul` of the entry points."""
        if getattr(self, "_split2", None) is None:
            Ws, bs, rinv = self.equilibrated()
            ws, meta, bp = [], [], []
            for W, b in zip(Ws, bs):
                sw = _pow2_weight_scale(W)
                ws.append(_pack_weight_split2(W * sw))
                # (max(b)_+: the bound feeds the NEXT layer's activation scale, and a post-ReLU value is bounded by the
                # positive part -- a row with a hugely negative bias does not loosen it)
                meta += [sw, float(W.abs().sum(1).max()), max(float(b.max()), 0.0)]
                bp.append(_pad32(b))
            self._split2 = (ws, (ctypes.c_void_p * self.n_layers)(*[t.data_ptr() for t in ws]),
                            (ctypes.c_float * len(meta))(*meta), bp,
                            (ctypes.c_void_p * self.n_layers)(*[t.data_ptr() for t in bp]), _pad32(rinv, fill=1.0))
        return self._split2[1], self._split2[2], self._split2[4], self._split2[5]

    def split(self):
        """-> ctypes array of the split-bf16 weight buffers (csrc/sa_mlp_split.hip), built on first use."""
        if self._equil_given is not None:
            raise RuntimeError("a rescaled (fp16 x 2) chain on a kernel that does not undo the rescaling")
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
    """The power of two that puts max|W| into (2^13, 2^14] (1 for a zero matrix; at most 2^60: the kernels' scale products
    stay finite, and a matrix that small is noise anyway)."""
    wmax = float(W.abs().max()) if W.numel() else 0.0
    if not wmax > 0:
        return 1.0
    sw = 2.0 ** (14 - math.ceil(math.log2(wmax)))
    while wmax * sw > 16384.0:
        sw *= 0.5
    return min(sw, 2.0 ** 60)


# What layer 0's input bound is taken to be when the rows of a chain are balanced on the host (the true bound lives on
# the device and only enters the kernels' scales; this number only decides how a row's weights weigh against its bias).
EQUIL_NOMINAL_INPUT = 1.0
EQUIL_MAX_BOOST = 8.0


def _pow2_inv(v):
    """Elementwise: the power of two r with r * v in (0.5, 1] (1 where v == 0 or not finite), clamped to [2^-60, 2^60]."""
    v = v.double()
    ok = torch.isfinite(v) & (v > 0)
    e = torch.ceil(torch.log2(torch.where(ok, v, torch.ones_like(v))))
    r = torch.pow(torch.full_like(v, 2.0), -e)
    r = torch.where(r * v > 1.0, r * 0.5, r)            # (log2 rounding at exact powers of two)
    r = torch.where(r * v <= 0.5, r * 2.0, r)
    r = torch.clamp(r, 2.0 ** -60, 2.0 ** 60)
    return torch.where(ok, r, torch.ones_like(r)).float()


def equilibrate(Ws, bs, b0=None):
    """Diagonal rescaling of a ReLU chain y_l = relu(W_l y_(l-1) + b_l) by powers of two (exact in fp32):
        W~_l = D_l W_l D_(l-1)^-1,  b~_l = D_l b_l,  D_(-1) = I,   so that   y~_l = D_l y_l   (ReLU commutes with D_l > 0)
    with D_l[o] = the power of two that puts channel o's upper bound u_o = (sum_k |W_l[o, k]| u_k + b_l[o])_+ into (0.5, 1]
    -- per-channel interval propagation from u = b0 (default EQUIL_NOMINAL_INPUT) on every input channel, the bias with
    its SIGN (a post-ReLU value is bounded by the positive part).  A channel the nominal input cannot switch on (u_o = 0,
    or below 2^-10 of its |W| u + |b|) is scaled by its weights alone and counts as 0 in the next layer's row bounds.
    Why: the fp16 x 2 kernels give a weight MATRIX one scale and a layer's activations one scale; a folded BatchNorm
    spreads the rows of W' = W gamma / sigma over orders of magnitude, and a row far below the matrix maximum loses its
    low fp16 piece to the subnormal range (measured: 5e-4 of the row's own output at a spread of 1e8).  After the
    rescaling every row of every layer sits within 2^9 of its matrix's maximum and every live hidden channel uses the
    full range of the activations' scale.  The last layer's D_L is undone on the outputs (rinv = 1 / D_L, the kernels'
    out_row_mul).  -> ([W~_l], [b~_l], rinv)."""
    b0 = EQUIL_NOMINAL_INPUT if b0 is None else float(b0)
    out_w, out_b = [], []
    r_prev = None
    u = torch.full((Ws[0].shape[1],), b0, dtype=torch.float64, device=Ws[0].device)      # bound per (scaled) input channel
    for W, b in zip(Ws, bs):
        Wc = W if r_prev is None else W / r_prev[None, :]
        lw = Wc.abs().double() @ u                                   # the weights' share of the row's bound
        ub = torch.clamp(lw + b.double(), min=0.0)
        live = ub > (lw + b.abs().double()) * 2.0 ** -10
        # (a row whose negative bias cancels most of its weights' range -- a channel that fires for rare inputs only -- is
        # not boosted by more than EQUIL_MAX_BOOST beyond the scale of its weights: the kernels' scalar bound
        # ||W~||_inf B + max(b~)_+ takes every row's weights at their worst, and a boosted row would loosen it for all)
        r = _pow2_inv(torch.maximum(torch.where(live, ub, lw), lw / EQUIL_MAX_BOOST))
        out_w.append((Wc * r[:, None]).contiguous())
        out_b.append((b * r).contiguous())
        u = torch.where(live, ub * r.double(), torch.zeros_like(ub))
        r_prev = r
    return out_w, out_b, (1.0 / r_prev).contiguous()


def _two_fp16(v):
    h = v.float().half().double()
    return h, (v - h).float().half().double()


def fp16x2_probe(Ws, bs, rinv=None, row_scaled=False, n=512, seed=0):
    """Per-output-channel error of the fp16 x 2 arithmetic on this chain beside the plain fp32 chain's, both against the
    chain's own float64 value, on a seeded probe (n inputs ~ N(0, EQUIL_NOMINAL_INPUT^2); always on the CPU: the answer
    must not depend on the device).  The two-piece side is the OPERAND error (float64 accumulation): per layer one
    power-of-two weight scale (row_scaled: one per row, the layer-by-layer GEMM's w_row_mul) with max|sw W| in (2^13, 2^14],
    one activation scale from the kernels' bound B' = ||W||_inf B + max(b)_+, operands as fp16(v) + fp16(v - fp16(v)),
    products wh.xh + wh.xl + wl.xh.  The fp32 side (torch's fp32 matmul) is the yardstick: a channel that cancels is as
    ill-conditioned there.  -> (e16, e32): (live channels,) tensors of max |err| / the channel's max |y|."""
    dev = torch.device("cpu")
    Ws = [W.detach().to(dev) for W in Ws]
    bs = [b.detach().to(dev) for b in bs]
    g = torch.Generator(device="cpu").manual_seed(seed)
    x0 = torch.randn(n, Ws[0].shape[1], generator=g, dtype=torch.float64) * EQUIL_NOMINAL_INPUT
    want, x32, x = x0, x0.float(), x0
    B = float(x0.abs().max())
    for W, b in zip(Ws, bs):
        Wd, bd = W.double(), b.double()
        want = torch.relu(want @ Wd.T + bd)
        x32 = torch.relu(x32 @ W.T + b)
        sx = math.ldexp(1.0, 14 - math.frexp(max(B, 1e-30))[1])
        if row_scaled:
            sw = (_pow2_inv(W.abs().amax(1)) * 16384.0).double()[:, None]
        else:
            sw = _pow2_weight_scale(W)
        wh, wl = _two_fp16(Wd * sw)
        xh, xl = _two_fp16(x * sx)
        acc = xh @ wh.T + xl @ wh.T + xh @ wl.T
        x = torch.relu(acc / (sw.T if row_scaled else sw) / sx + bd)
        B = (float(W.abs().sum(1).max()) * B + max(float(b.max()), 0.0)) * 1.01
    sc = want.abs().amax(0)
    live = sc > 0
    e16 = ((x - want).abs().amax(0) / sc.clamp_min(1e-300))[live]
    e32 = ((x32.double() - want).abs().amax(0) / sc.clamp_min(1e-300))[live]
    return e16, e32


# fp16 x 2 is taken for a chain when, on the probe, its worst channel is below FP16X2_PROBE_BAR (what an fp32 FMA chain
# leaves on a well-conditioned channel) or within FP16X2_PROBE_FACTOR of the fp32 chain's own worst channel, AND nine
# channels in ten are within FP16X2_PROBE_Q90 of the fp32 chain's error on the same channel (range trouble -- a low piece
# in fp16's subnormal range -- hits every channel that reads the affected operand).  Otherwise the chain runs bf16 x 3
# (8 exponent bits per piece: no range to manage).
# Short contractions: two fp16 pieces carry 22 bits of each operand, an fp32 product 24; with hundreds of terms the fp32
# chain's accumulation roundings dominate and the two arithmetics end up equally far from float64 (measured: 0.3-0.5 x),
# with the 9 .. 32 terms of SA level 0 they do not, and on a channel that cancels the two-piece result is 2-4 x the fp32
# chain's (still 1e-6 .. 1e-5 of the channel's scale: the arithmetic's nominal accuracy, not range trouble).
FP16X2_PROBE_BAR = 2.0e-6
FP16X2_PROBE_FACTOR = 3.0
FP16X2_PROBE_Q90 = 2.0
FP16X2_PROBE_SHORT_K = 64             # first-layer contractions shorter than this: factor and quantile bar x 2


def fp16x2_verdict(e16, e32, k_first=1 << 30):
    """-> (safe, worst e16, worst e32, 90 % quantile of e16 / max(e32, 5e-7)) from fp16x2_probe's tensors (the floor: two
    errors that are both below a quarter of the bar are not compared -- with nine input channels the fp32 chain rounds a
    handful of times and the two pieces' own 2^-22 per operand is the larger of two negligible numbers)."""
    if e16.numel() == 0:
        return True, 0.0, 0.0, 0.0
    q90 = float(torch.quantile(e16 / e32.clamp_min(0.25 * FP16X2_PROBE_BAR), 0.9))
    m16, m32 = float(e16.max()), float(e32.max())
    short = 2.0 if k_first < FP16X2_PROBE_SHORT_K else 1.0
    return (m16 <= max(FP16X2_PROBE_BAR, short * FP16X2_PROBE_FACTOR * m32) and q90 <= short * FP16X2_PROBE_Q90), m16, m32, q90


def _pad32(v, fill=0.0):
    out = torch.full((((v.numel() + 31) // 32) * 32,), float(fill), dtype=torch.float32, device=v.device)
    out[:v.numel()] = v
    return out


def _pack_weight_h16_rows(W, slabs):
    """W (M, K) float32 -> (h16 buffer as _pack_weight_h16 of the ROW-scaled matrix, device float32[roundup128(M)] row
    multipliers): row o is stored as rs[o] W[o] with rs[o] the power of two that puts the row's maximum into (2^13, 2^14]
    and the multiplier 1 / rs[o] is what pvn3d_split_gemm2's w_row_mul undoes on the accumulators (w_scale = 1)."""
    M = W.shape[0]
    rs = _pow2_inv(W.abs().amax(1) if W.shape[1] else torch.zeros(M, device=W.device)) * 16384.0
    rs = torch.clamp(rs, max=2.0 ** 60)
    rm = torch.ones(((M + 127) // 128) * 128, dtype=torch.float32, device=W.device)
    rm[:M] = 1.0 / rs
    return _pack_weight_h16(W * rs[:, None], slabs), rm


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
