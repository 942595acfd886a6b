"""Training-mode SharedMLP on the hand-written bf16 MFMA kernels (csrc/mlp_train.hip).

Replaces, when a set-abstraction / feature-propagation module is in training mode on the GPU, the reference's
    grouped (B, C, npoint, nsample) tensor -> [Conv2d 1x1 -> BatchNorm2d(batch stats) -> ReLU] x L -> max_pool2d
(pvn3d/lib/pointnet2_utils/pointnet2_modules.py:58-71, 188-206; pvn3d/lib/utils/etw_pytorch_utils/
pytorch_utils.py:25-50) and its autograd backward -- cuDNN / MIOpen convolutions, BatchNorm and pooling kernels on
NCHW tensors in the reference -- by point-major bf16 matrices, one GEMM kernel and a handful of fused
elementwise / reduction kernels (see the header of csrc/mlp_train.hip).  Parameters stay the module's own fp32
tensors (state_dict unchanged); BatchNorm running statistics and num_batches_tracked are updated like
nn.BatchNorm2d does in training mode.

Numerics: bf16 activations and weights, fp32 accumulation, fp32 statistics -- the precision of the reference
training step under ``torch.autocast(dtype=torch.bfloat16)`` (BASELINE config 5).
"""
import torch
import torch.nn as nn

from ..._lib import lib, check, on_device
from . import _ext

# Which SharedMLP path SA / FP modules take in training mode:
#   "auto" (default): the bf16 chain of this file only under ``torch.autocast(device_type="cuda", dtype=torch.bfloat16)``
#            -- the caller asked for bf16 arithmetic; plain fp32 training (what the reference's scripts do: no AMP in
#            train_linemod_pvn3d.py:169-212,375) keeps fp32 numerics;
#   True:    explicit opt-in, the bf16 chain whatever the autocast state;
#   False:   always the reference's op-by-op composition (torch Conv2d / BatchNorm2d).
TRAIN_FUSED = "auto"


def train_fused_enabled():
    """-> whether a module in training mode should take the bf16 MFMA chain right now (see TRAIN_FUSED)."""
    if TRAIN_FUSED == "auto":
        return torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
    return bool(TRAIN_FUSED)


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _bump_version(t):
    """Advance t._version without launching a kernel (the data was written by a raw-pointer kernel)."""
    torch._C._autograd._unsafe_set_version_counter([t], [t._version + 1])


# kernel limits of csrc/mlp_train.hip (pool / BatchNorm passes): beyond them the modules keep the op-by-op composition
MAX_NSAMPLE = 256
MAX_LD = 2048


def _ld(c):
    return (int(c) + 15) // 16 * 16


def _strides3(t):
    """(B, C, N) float32 tensor with arbitrary strides -> (data_ptr, sb, sc, sn)."""
    return t.data_ptr(), t.stride(0), t.stride(1), t.stride(2)


def shared_mlp_layers(mlp):
    """SharedMLP -> [(conv, bn)] when every layer is exactly [1x1 Conv2d without bias] -> BatchNorm2d -> ReLU
    (the only form PVN3D builds) with the BatchNorm in training mode (a frozen / eval BatchNorm inside a module that
    is in training mode normalises with its running statistics and must not update them: that composition is left
    to torch); None otherwise."""
    out = []
    for layer in mlp.children():
        names = [n for n, _ in layer.named_children()]
        conv = getattr(layer, "conv", None)
        norm = getattr(layer, "normlayer", None)
        bn = getattr(norm, "bn", None) if norm is not None else None
        act = getattr(layer, "activation", None)
        if (names[:1] != ["conv"] or not isinstance(conv, nn.Conv2d) or conv.kernel_size != (1, 1)
                or conv.stride != (1, 1) or conv.padding != (0, 0) or conv.groups != 1 or conv.bias is not None
                or not isinstance(bn, nn.BatchNorm2d) or not bn.affine or not bn.track_running_stats
                or bn.momentum is None or not bn.training or not isinstance(act, nn.ReLU)):
            return None
        out.append((conv, bn))
    return out or None


class _Chain(object):
    """Forward / backward of one [GEMM -> BatchNorm(batch stats) -> ReLU] x L chain on a point-major bf16 input
    X0 (rows, ld0).  Holds what the backward needs."""

    def __init__(self, x0, c_in, weights, gammas, betas, bns):
        self.dev = x0.device
        self.rows = x0.size(0)
        self.x0 = x0
        self.c = [int(c_in)] + [int(w.size(0)) for w in weights]
        self.weights, self.gammas, self.betas, self.bns = weights, gammas, betas, bns
        self.y, self.h, self.mean, self.invstd, self.a = [], [], [], [], []

    def forward(self, update_running=True, pool=None):
        """pool = (G, ns, out_ptr, out_ld, arg): the last layer's ReLU output goes straight into the max-pool (its
        matrix is not materialised); returns None then."""
        dev, rows, st = self.dev, self.rows, None
        prev = self.x0
        L = len(self.weights)
        for li, w in enumerate(self.weights):
            cin, cout = self.c[li], self.c[li + 1]
            ldi, ldo = _ld(cin), _ld(cout)
            st = _stream(prev)
            wb = torch.empty((cout, ldi), dtype=torch.bfloat16, device=dev)
            check(lib.pvn3d_mt_pack_weight(cout, cin, w.data_ptr(), cin, 0, wb.data_ptr(), cout, ldi, st), "mt_pack_weight")
            y = torch.empty((rows, ldo), dtype=torch.bfloat16, device=dev)
            P = lib.pvn3d_mt_gemm_nt_stat_rows(rows)
            ps = torch.empty((2, P, ldo), dtype=torch.float32, device=dev)
            check(lib.pvn3d_mt_gemm_nt(rows, cout, ldi, prev.data_ptr(), ldi, wb.data_ptr(), ldi, y.data_ptr(), ldo,
                                       ps[0].data_ptr(), ps[1].data_ptr(), ldo, st), "mt_gemm_nt")
            bn = self.bns[li]
            stats = torch.empty((4, ldo), dtype=torch.float32, device=dev)      # mean, invstd, a, b
            track = update_running and bn.running_mean is not None
            check(lib.pvn3d_mt_bn_finalize(P, ldo, cout, float(rows), ps[0].data_ptr(), ps[1].data_ptr(),
                                           self.gammas[li].data_ptr(), self.betas[li].data_ptr(), float(bn.eps),
                                           float(bn.momentum), bn.running_mean.data_ptr() if track else None,
                                           bn.running_var.data_ptr() if track else None, stats[0].data_ptr(),
                                           stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), st),
                  "mt_bn_finalize")
            if track:
                # the kernel wrote the running statistics through raw pointers: tell everything that caches on
                # (data_ptr, _version) -- the folded eval weights of the fused inference kernels -- that they changed
                _bump_version(bn.running_mean)
                _bump_version(bn.running_var)
                if bn.num_batches_tracked is not None:
                    bn.num_batches_tracked += 1
            self.y.append(y)
            self.mean.append(stats)
            if pool is not None and li == L - 1:
                G, ns, out_ptr, out_ld, arg = pool
                check(lib.pvn3d_mt_bn_relu_pool(G, ns, ldo, cout, y.data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(),
                                                out_ptr, out_ld, arg.data_ptr(), st), "mt_bn_relu_pool")
                self.h.append(None)
                return None
            h = torch.empty((rows, ldo), dtype=torch.bfloat16, device=dev)
            check(lib.pvn3d_mt_bn_relu_apply(rows, ldo, y.data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(),
                                             h.data_ptr(), st), "mt_bn_relu_apply")
            self.h.append(h)
            prev = h
        return prev

    def backward(self, dh, need_input_grad, pooled=None):
        """dh: bf16 (rows, ld_L) gradient w.r.t. the last layer's output, or None with pooled = (G, ns, dout_ptr,
        out_ld, arg): that gradient is the max-pool backward of dout and is never materialised.  Returns (dX0 or
        None, [dW], [dgamma], [dbeta])."""
        dev, rows = self.dev, self.rows
        st = _stream(self.x0)
        L = len(self.weights)
        dws, dgs, dbs = [None] * L, [None] * L, [None] * L
        for li in range(L - 1, -1, -1):
            cin, cout = self.c[li], self.c[li + 1]
            ldi, ldo = _ld(cin), _ld(cout)
            y, stats = self.y[li], self.mean[li]
            use_pool = pooled is not None and li == L - 1
            if use_pool:
                G, ns, dout_ptr, out_ld, arg = pooled
                P = lib.pvn3d_mt_bn_bwd_partials(G)
                pp = torch.empty((2, P, ldo), dtype=torch.float32, device=dev)
                check(lib.pvn3d_mt_bn_bwd_reduce_pooled(G, ns, ldo, cout, dout_ptr, out_ld, arg.data_ptr(), y.data_ptr(),
                                                        stats[2].data_ptr(), stats[3].data_ptr(), stats[0].data_ptr(),
                                                        stats[1].data_ptr(), pp[0].data_ptr(), pp[1].data_ptr(), st),
                      "mt_bn_bwd_reduce_pooled")
            else:
                P = lib.pvn3d_mt_bn_bwd_partials(rows)
                pp = torch.empty((2, P, ldo), dtype=torch.float32, device=dev)
                check(lib.pvn3d_mt_bn_bwd_reduce(rows, ldo, dh.data_ptr(), y.data_ptr(), stats[2].data_ptr(),
                                                 stats[3].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                                                 pp[0].data_ptr(), pp[1].data_ptr(), st), "mt_bn_bwd_reduce")
            dgb = torch.empty((2, cout), dtype=torch.float32, device=dev)
            kk = torch.empty((2, ldo), dtype=torch.float32, device=dev)
            check(lib.pvn3d_mt_bn_bwd_finalize(P, ldo, cout, float(rows), pp[0].data_ptr(), pp[1].data_ptr(),
                                               stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(),
                                               dgb[0].data_ptr(), dgb[1].data_ptr(), kk[0].data_ptr(), kk[1].data_ptr(),
                                               st), "mt_bn_bwd_finalize")
            dy = torch.empty((rows, ldo), dtype=torch.bfloat16, device=dev)
            if use_pool:
                check(lib.pvn3d_mt_bn_bwd_apply_pooled(G, ns, ldo, cout, dout_ptr, out_ld, arg.data_ptr(), y.data_ptr(),
                                                       stats[2].data_ptr(), stats[3].data_ptr(), kk[0].data_ptr(),
                                                       kk[1].data_ptr(), dy.data_ptr(), st), "mt_bn_bwd_apply_pooled")
            else:
                check(lib.pvn3d_mt_bn_bwd_apply(rows, ldo, dh.data_ptr(), y.data_ptr(), stats[2].data_ptr(),
                                                stats[3].data_ptr(), kk[0].data_ptr(), kk[1].data_ptr(), dy.data_ptr(), st),
                      "mt_bn_bwd_apply")
            dgs[li], dbs[li] = dgb[0], dgb[1]
            # weight gradient: dW (cout, cin) = dY^T . H_prev, K = rows
            prev = self.h[li - 1] if li > 0 else self.x0
            dw = torch.zeros((cout, cin), dtype=torch.float32, device=dev)
            if WGRAD_TN and lib.pvn3d_mt_wgrad_tn_ok(cout, cin):
                # straight from the row-major matrices (no transposed copies)
                check(lib.pvn3d_mt_wgrad_tn(rows, cout, cin, dy.data_ptr(), ldo, prev.data_ptr(), ldi, dw.data_ptr(), cin,
                                            st), "mt_wgrad_tn")
            else:
                # the transposed copies hold the contraction dimension (rows) contiguously: it has to be a multiple of
                # 16 (8 rows per 16-byte store of the transpose, K % 16 of the GEMM) -- zero rows add nothing to dW
                r16 = (rows + 15) // 16 * 16
                dy_s, pv_s = dy, prev
                if r16 != rows:
                    dy_s = torch.zeros((r16, ldo), dtype=torch.bfloat16, device=dev)
                    dy_s[:rows] = dy
                    pv_s = torch.zeros((r16, ldi), dtype=torch.bfloat16, device=dev)
                    pv_s[:rows] = prev
                dyt = torch.empty((ldo, r16), dtype=torch.bfloat16, device=dev)
                pvt = torch.empty((ldi, r16), dtype=torch.bfloat16, device=dev)
                check(lib.pvn3d_mt_transpose(r16, ldo, dy_s.data_ptr(), dyt.data_ptr(), r16, st), "mt_transpose")
                check(lib.pvn3d_mt_transpose(r16, ldi, pv_s.data_ptr(), pvt.data_ptr(), r16, st), "mt_transpose")
                tiles = ((cout + 127) // 128) * ((cin + 127) // 128)
                ksplit = max(1, min(r16 // 512, 1024 // tiles))
                check(lib.pvn3d_mt_gemm_nt_splitk(cout, cin, r16, dyt.data_ptr(), r16, pvt.data_ptr(), r16,
                                                  dw.data_ptr(), cin, ksplit, st), "mt_gemm_nt_splitk")
                del dyt, pvt, dy_s, pv_s
            dws[li] = dw
            # input gradient: dH_prev (rows, ldi) = dY . W
            if li > 0 or need_input_grad:
                wt = torch.empty((cin, ldo), dtype=torch.bfloat16, device=dev)
                check(lib.pvn3d_mt_pack_weight(cout, cin, self.weights[li].data_ptr(), cin, 1, wt.data_ptr(), cin, ldo,
                                               st), "mt_pack_weight")
                dprev = torch.empty((rows, ldi), dtype=torch.bfloat16, device=dev)
                check(lib.pvn3d_mt_gemm_nt(rows, cin, ldo, dy.data_ptr(), ldo, wt.data_ptr(), ldo, dprev.data_ptr(), ldi,
                                           None, None, 0, st), "mt_gemm_nt")
                dh = dprev
            else:
                dh = None
        return dh, dws, dgs, dbs


# Backward of the layer-0 gathers through inverted index lists (csrc/mlp_train.hip mt_csr_build / mt_inv_gather)
# instead of the atomic scatter kernels; False restores mt_unpack_cm + group_points_grad / three_interpolate_grad.
INVERSE_GATHER = True
# Weight gradients straight from the row-major matrices (mt_wgrad_tn; layers of up to 512 x 544 channels);
# False: transposed copies + the split-K NT GEMM for every layer.
WGRAD_TN = True


def _inv_gather(dx0, B, n_src, idx, div, C, c_off, w, out, accumulate, st):
    """out (B, n_src, C) fp32 (+)= scatter-add of dx0's channels [c_off, c_off + C) along idx, as a gather."""
    dev = dx0.device
    E = idx.numel() // B
    start = torch.empty((B, n_src + 1), dtype=torch.int32, device=dev)
    ent = torch.empty((B, E), dtype=torch.int32, device=dev)
    check(lib.pvn3d_mt_csr_build(B, n_src, E, idx.data_ptr(), start.data_ptr(), ent.data_ptr(), st), "mt_csr_build")
    for c0 in range(0, C, 512):             # channel blocks of <= 512
        cb = min(512, C - c0)
        check(lib.pvn3d_mt_inv_gather(B, n_src, E, div, cb, c_off + c0, dx0.size(1), dx0.data_ptr(), start.data_ptr(),
                                      ent.data_ptr(), w.data_ptr() if w is not None else None,
                                      out.data_ptr() + 4 * c0, C, 1 if accumulate else 0, st), "mt_inv_gather")


def _flat_params(layer_lists):
    """[[(conv, bn)]] -> flat tensor list (conv.weight, bn.weight, bn.bias per layer) for autograd."""
    flat = []
    for layers in layer_lists:
        for conv, bn in layers:
            flat += [conv.weight, bn.weight, bn.bias]
    return flat


class SALevelTrain(torch.autograd.Function):
    """One multi-scale set-abstraction level in training mode: per scale gather -> MLP chain -> max-pool, every
    scale writing its channel slice of ONE point-major (B, npoint, C_total) fp32 tensor (what the reference
    builds with group_points + cat + SharedMLP + max_pool2d + cat)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, spec, *params):
        # spec: list per scale of (idx (B, npoint, ns) int32, use_xyz, [(conv, bn)])
        B, N, m = xyz.size(0), xyz.size(1), new_xyz.size(1)
        dev = xyz.device
        C = features.size(1) if features is not None else 0
        widths = [layers[-1][0].out_channels for _, _, layers in spec]
        total = sum(widths)
        out = torch.empty((B, m, total), dtype=torch.float32, device=dev)
        feats = features.detach() if features is not None else None
        chains, args, off, p = [], [], 0, 0
        with on_device(dev):
            for (idx, use_xyz, layers), width in zip(spec, widths):
                L = len(layers)
                ws = [params[p + 3 * i].detach().view(layers[i][0].out_channels, -1) for i in range(L)]
                gs = [params[p + 3 * i + 1].detach() for i in range(L)]
                bs = [params[p + 3 * i + 2].detach() for i in range(L)]
                p += 3 * L
                ns = idx.size(2)
                nx = 3 if use_xyz else 0
                c0 = nx + C
                rows = B * m * ns
                x0 = torch.empty((rows, _ld(c0)), dtype=torch.bfloat16, device=dev)
                fp, fsb, fsc, fsn = _strides3(feats) if feats is not None else (None, 0, 0, 0)
                check(lib.pvn3d_mt_gather_sa(B, N, m, ns, C, 1 if use_xyz else 0, xyz.data_ptr(), new_xyz.data_ptr(), fp,
                                             fsb, fsc, fsn, idx.data_ptr(), x0.data_ptr(), _ld(c0), _stream(xyz)),
                      "mt_gather_sa")
                ch = _Chain(x0, c0, ws, gs, bs, [bn for _, bn in layers])
                arg = torch.empty((B * m, _ld(width)), dtype=torch.uint8, device=dev)
                ch.forward(pool=(B * m, ns, out.data_ptr() + 4 * off, total, arg))
                chains.append(ch)
                args.append((arg, idx, use_xyz, ns, width, off))
                off += width
        ctx.chains, ctx.args = chains, args
        ctx.shape = (B, N, m, C, total)
        ctx.feat_meta = (features.shape, features.requires_grad) if features is not None else (None, False)
        return out

    @staticmethod
    def backward(ctx, gout):
        B, N, m, C, total = ctx.shape
        if ctx.chains is None:
            raise RuntimeError("the fused training chain frees its saved activations in backward: a second backward "
                               "through the same graph (retain_graph=True) is not supported; set "
                               "_train_mlp.TRAIN_FUSED = False for that")
        gout = gout.contiguous()
        dev = gout.device
        fshape, fneeds = ctx.feat_meta
        dfeat = None
        grads = []
        point_major = fneeds and INVERSE_GATHER and N <= 32768
        with on_device(dev):
            st = _stream(gout)
            for ch, (arg, idx, use_xyz, ns, width, off) in zip(ctx.chains, ctx.args):
                dx0, dws, dgs, dbs = ch.backward(None, fneeds, pooled=(B * m, ns, gout.data_ptr() + 4 * off, total, arg))
                if point_major:
                    # point-major (B, N, C) gradient, handed back as the transposed view the previous level produced
                    first = dfeat is None
                    if first:
                        dfeat = torch.empty((B, N, C), dtype=torch.float32, device=dev)
                    _inv_gather(dx0, B, N, idx, 1, C, 3 if use_xyz else 0, None, dfeat, not first, st)
                elif fneeds:
                    # feature channels of dX0 as (B, C, npoint, nsample) fp32 -> the row-owner scatter of group_points_grad
                    gcm = torch.empty((B, C, m, ns), dtype=torch.float32, device=dev)
                    check(lib.pvn3d_mt_unpack_cm(B, m * ns, dx0.size(1), 3 if use_xyz else 0, C, dx0.data_ptr(),
                                                 gcm.data_ptr(), st), "mt_unpack_cm")
                    part = _ext.group_points_grad(gcm, idx, N)
                    dfeat = part if dfeat is None else dfeat + part
                for li, dw in enumerate(dws):
                    grads += [dw.view(dw.size(0), dw.size(1), 1, 1), dgs[li], dbs[li]]
        ctx.chains = None
        if dfeat is not None and point_major:
            dfeat = dfeat.transpose(1, 2)          # (B, C, N) view of the point-major buffer
        return (None, None, dfeat, None) + tuple(grads)


class FPTrain(torch.autograd.Function):
    """One feature-propagation module in training mode: three_interpolate ++ skip features -> MLP chain; returns
    the point-major (B, n, C_out) fp32 tensor."""

    @staticmethod
    def forward(ctx, unknow_feats, known_feats, idx, weight, layers, channel_major, *params):
        B, C2, mk = known_feats.shape
        n = idx.size(1)
        C1 = unknow_feats.size(1) if unknow_feats is not None else 0
        dev = known_feats.device
        L = len(layers)
        ws = [params[3 * i].detach().view(layers[i][0].out_channels, -1) for i in range(L)]
        gs = [params[3 * i + 1].detach() for i in range(L)]
        bs = [params[3 * i + 2].detach() for i in range(L)]
        c0 = C2 + C1
        rows = B * n
        kf = known_feats.detach()
        uf = unknow_feats.detach() if unknow_feats is not None else None
        with on_device(dev):
            st = _stream(known_feats)
            x0 = torch.empty((rows, _ld(c0)), dtype=torch.bfloat16, device=dev)
            kp, ksb, ksc, ksn = _strides3(kf)
            up, usb, usc, usn = _strides3(uf) if uf is not None else (None, 0, 0, 0)
            check(lib.pvn3d_mt_gather_fp(B, n, mk, C2, C1, kp, ksb, ksc, ksn, up, usb, usc, usn, idx.data_ptr(),
                                         weight.data_ptr(), x0.data_ptr(), _ld(c0), st), "mt_gather_fp")
            ch = _Chain(x0, c0, ws, gs, bs, [bn for _, bn in layers])
            h = ch.forward()
            cout = ch.c[-1]
            if channel_major:        # the reference's contiguous (B, C_out, n)
                out = torch.empty((B, cout, n), dtype=torch.float32, device=dev)
                check(lib.pvn3d_mt_unpack_cm(B, n, _ld(cout), 0, cout, h.data_ptr(), out.data_ptr(), st), "mt_unpack_cm")
            else:                    # point-major (B, n, C_out): the next level gathers rows
                out = torch.empty((B, n, cout), dtype=torch.float32, device=dev)
                check(lib.pvn3d_mt_unpack_out(rows, _ld(cout), cout, h.data_ptr(), out.data_ptr(), cout, st),
                      "mt_unpack_out")
        ctx.channel_major = channel_major
        ctx.chain = ch
        ctx.meta = (B, n, mk, C2, C1, idx, weight,
                    unknow_feats is not None and unknow_feats.requires_grad, known_feats.requires_grad)
        return out

    @staticmethod
    def backward(ctx, gout):
        B, n, mk, C2, C1, idx, weight, need_u, need_k = ctx.meta
        ch = ctx.chain
        if ch is None:
            raise RuntimeError("the fused training chain frees its saved activations in backward: a second backward "
                               "through the same graph (retain_graph=True) is not supported; set "
                               "_train_mlp.TRAIN_FUSED = False for that")
        gout = gout.contiguous()
        dev = gout.device
        cout = ch.c[-1]
        with on_device(dev):
            st = _stream(gout)
            dh = torch.empty((ch.rows, _ld(cout)), dtype=torch.bfloat16, device=dev)
            if ctx.channel_major:
                check(lib.pvn3d_mt_pack_cm(B, n, _ld(cout), cout, gout.data_ptr(), dh.data_ptr(), st), "mt_pack_cm")
            else:
                check(lib.pvn3d_mt_pack_grad(ch.rows, _ld(cout), cout, gout.data_ptr(), cout, dh.data_ptr(), st),
                      "mt_pack_grad")
            dx0, dws, dgs, dbs = ch.backward(dh, need_u or need_k)
            dk = du = None
            if need_k and INVERSE_GATHER and mk <= 32768:
                dk = torch.empty((B, mk, C2), dtype=torch.float32, device=dev)
                _inv_gather(dx0, B, mk, idx, 3, C2, 0, weight, dk, False, st)
                dk = dk.transpose(1, 2)
            elif need_k:
                gk = torch.empty((B, C2, n), dtype=torch.float32, device=dev)
                check(lib.pvn3d_mt_unpack_cm(B, n, dx0.size(1), 0, C2, dx0.data_ptr(), gk.data_ptr(), st), "mt_unpack_cm")
                dk = _ext.three_interpolate_grad(gk, idx, weight, mk)
            if need_u:
                du = torch.empty((B, C1, n), dtype=torch.float32, device=dev)
                check(lib.pvn3d_mt_unpack_cm(B, n, dx0.size(1), C2, C1, dx0.data_ptr(), du.data_ptr(), st), "mt_unpack_cm")
        grads = []
        for li, dw in enumerate(dws):
            grads += [dw.view(dw.size(0), dw.size(1), 1, 1), dgs[li], dbs[li]]
        ctx.chain = None
        return (du, dk, None, None, None, None) + tuple(grads)


def sa_level_train(module, xyz, new_xyz, features, idxs):
    """PointnetSAModuleMSG.forward in training mode -> (B, C_total, npoint) (a transposed view of the point-major
    result), or None when a scale is not of the supported form."""
    from . import pointnet2_utils
    if xyz.requires_grad or new_xyz.requires_grad:
        return None          # the chain has no gradient w.r.t. the coordinates; the op-by-op composition does
    spec, layer_lists = [], []
    c_feat = features.size(1) if features is not None else 0
    for grouper, mlp, idx in zip(module.groupers, module.mlps, idxs):
        layers = shared_mlp_layers(mlp)
        if layers is None or not isinstance(grouper, pointnet2_utils.QueryAndGroup) or idx is None:
            return None
        if idx.size(2) > MAX_NSAMPLE or max([c_feat + 3] + [conv.out_channels for conv, _ in layers]) > MAX_LD - 16:
            return None
        if features is None and not grouper.use_xyz:
            return None
        use_xyz = grouper.use_xyz or features is None
        spec.append((idx, use_xyz, layers))
        layer_lists.append(layers)
    out = SALevelTrain.apply(xyz, new_xyz, features, spec, *_flat_params(layer_lists)).transpose(1, 2)
    # stand-alone the module returns the reference's contiguous (B, C, npoint); inside Pointnet2MSG the next level
    # gathers rows from the point-major buffer, so the transposed view is handed over
    return out if getattr(module, "_point_major_out", False) else out.contiguous()


def fp_train(module, unknow_feats, known_feats, idx, weight):
    """PointnetFPModule.forward in training mode -> (B, C_out, n) (transposed view), or None."""
    layers = shared_mlp_layers(module.mlp)
    if layers is None:
        return None
    c_in = known_feats.size(1) + (unknow_feats.size(1) if unknow_feats is not None else 0)
    if max([c_in] + [conv.out_channels for conv, _ in layers]) > MAX_LD - 16:
        return None
    channel_major = not getattr(module, "_point_major_out", False)
    out = FPTrain.apply(unknow_feats, known_feats, idx.contiguous(), weight.contiguous(), layers, channel_major,
                        *_flat_params([layers]))
    return out if channel_major else out.transpose(1, 2)
