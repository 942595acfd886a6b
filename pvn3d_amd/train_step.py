"""One data-parallel training step of the PointNet++ voting branch (BASELINE config 5).

What runs where:
  * FPS, gather, ball query, grouping, three_nn, three_interpolate and ALL their gradients: the hand-written
    gfx950 kernels of this package (fp32 gathers / atomic or deterministic scatters), through the reference's
    autograd Function API (lib/pointnet2_utils/pointnet2_utils.py);
  * the SharedMLP layers of every SA / FP module (the grouped-point x MLP-weight contraction: 1x1 conv ->
    BatchNorm with batch statistics -> ReLU, then max-pool) forward and backward: csrc/mlp_train.hip through
    lib/pointnet2_utils/_train_mlp.py -- point-major bf16 activations, one hand-written bf16 MFMA GEMM kernel
    (fp32 accumulate; BatchNorm statistics in its epilogue; split-K for the weight gradient) and fused
    BatchNorm / ReLU / pool kernels, fp32 master weights -- taken under ``autocast(bfloat16)`` only (or with the
    explicit opt-in ``_train_mlp.TRAIN_FUSED = True``); a plain fp32 step, like the reference's, keeps fp32
    numerics on torch Conv2d / BatchNorm2d (``_train_mlp.TRAIN_FUSED = False`` forces that path, = torch
    Conv2d / BatchNorm2d = MIOpen / hipBLASLt);
  * the two small per-point heads of this file (not part of the reference's hot path): torch Conv1d / BatchNorm1d
    under ``torch.autocast``;
  * the vote loss (of_l1_loss, lib/loss.py) forward and backward: csrc/vote_loss.hip, fp32;
  * gradient averaging across ranks: bucketed asynchronous all-reduce issued from inside backward
    (sharding.OverlappedGradientReducer: post-accumulate-grad hooks; RCCL over xGMI) instead of the reference's
    nn.DataParallel reduce-to-GPU-0 (train_linemod_pvn3d.py:480).

The reference's RGB CNN, DenseFusion and segmentation head (lib/pvn3d.py:157-322) are out of scope; the
per-point keypoint / centre offset heads here are the minimal 1x1-conv heads needed to drive the backward of
the hot path with the reference's own loss (train_linemod_pvn3d.py:307-375: loss_kp_of + loss_ctr_of).
"""
import torch
import torch.nn as nn

from . import sharding
from .lib.loss import OFLoss
from .lib.pointnet2_utils import _fused_mlp
from .lib.pointnet2_msg import Pointnet2MSG


class PointVoteNet(nn.Module):
    """Pointnet2MSG backbone + per-point keypoint-offset and centre-offset heads.
    forward(pointcloud (B, N, 3 + C)) -> pred_kp_of (B, K, N, 3), pred_ctr_of (B, 1, N, 3)."""

    def __init__(self, input_channels=6, n_kps=8, width=128):
        super(PointVoteNet, self).__init__()
        self.n_kps = n_kps
        self.backbone = Pointnet2MSG(input_channels=input_channels)
        self.kp_head = nn.Sequential(nn.Conv1d(128, width, 1), nn.BatchNorm1d(width), nn.ReLU(inplace=True),
                                     nn.Conv1d(width, n_kps * 3, 1))
        self.ctr_head = nn.Sequential(nn.Conv1d(128, width, 1), nn.BatchNorm1d(width), nn.ReLU(inplace=True),
                                      nn.Conv1d(width, 3, 1))

    def forward(self, pointcloud, geometry=None):
        feats = self.backbone(pointcloud, geometry=geometry)   # (B, 128, N)
        B, _, N = feats.shape
        kp = self.kp_head(feats).view(B, self.n_kps, 3, N).permute(0, 1, 3, 2).contiguous()
        ctr = self.ctr_head(feats).view(B, 1, 3, N).permute(0, 1, 3, 2).contiguous()
        return kp, ctr


def vote_loss(pred_kp_of, pred_ctr_of, kp_targ_ofst, ctr_targ_ofst, labels):
    """loss_kp_of + loss_ctr_of of the reference's model_fn (train_linemod_pvn3d.py:327-336), fp32."""
    crit = OFLoss()
    return crit(pred_kp_of, kp_targ_ofst, labels).sum() + crit(pred_ctr_of, ctr_targ_ofst, labels).sum()


def train_step(model, optimizer, batch, autocast_dtype=None, group=None, bucket_bytes=4 << 20, prefetch=None, exchange=True):
    """forward + vote loss + backward + gradient all-reduce + optimizer step.  batch: dict(pc (B,N,3+C),
    kp_targ_ofst (B,N,K,3), ctr_targ_ofst (B,N,1,3), labels (B,N,1)).  Returns the (detached) loss.
    exchange=False: no gradient exchange at all, whatever process group exists (the baseline leg of bench.py's
    distributed training entry; group=None alone means "the default group", which still exchanges).
    bucket_bytes: gradient all-reduce bucket size.  The voting branch has 14 MB of fp32 gradients, so 4 MiB gives four
    buckets to overlap with backward (the FP levels' -- produced first -- fly while the SA levels are computed); with
    the reference's full model (CNN included: ~160 MB) 32-64 MiB buckets amortise the per-collective latency better.

    prefetch: the NEXT step's point-cloud tensor (the data loader has it while this step runs).  Its xyz-only work --
    furthest point sampling, ball queries, three_nn: no parameters involved -- is enqueued on the geometry stream
    under this step's backward (FPS is one workgroup per cloud: 1.5 ms during which most of the chip idles) and the
    next call picks the handle up when it is handed the same tensor, unmodified."""
    model.train()
    optimizer.zero_grad(set_to_none=True)
    pc = batch["pc"]
    geo = None
    net = getattr(model, "module", model)        # a DistributedDataParallel / DataParallel wrapper holds the real model
    held = getattr(net, "_geometry_prefetched", None)
    if held is not None:
        net._geometry_prefetched = None
        # the handle belongs to ONE tensor object (kept alive here, so its address cannot be handed to another cloud
        # by the caching allocator) in the state it had when the geometry was computed
        if held[0] is pc and held[1] == pc._version:
            geo = held[2]
    if autocast_dtype is not None:
        with torch.autocast(device_type="cuda", dtype=autocast_dtype):
            kp, ctr = model(pc, geometry=geo)
            loss = vote_loss(kp, ctr, batch["kp_targ_ofst"], batch["ctr_targ_ofst"], batch["labels"])
    else:
        kp, ctr = model(pc, geometry=geo)
        loss = vote_loss(kp, ctr, batch["kp_targ_ofst"], batch["ctr_targ_ofst"], batch["labels"])
    if prefetch is not None and prefetch.is_cuda:
        with torch.no_grad():
            net._geometry_prefetched = (prefetch, prefetch._version, net.backbone.geometry_ahead(prefetch))
    # gradient buckets are all-reduced from inside backward (post-accumulate-grad hooks): the late layers' buckets are
    # on the wire while the early layers' gradients are still being computed (None: a single process, nothing to do)
    reducer = sharding.overlapped_reducer(net, bucket_bytes=bucket_bytes, group=group) if exchange else None
    if reducer is not None:
        reducer.arm()                    # the hooks act for this backward only
    try:
        loss.backward()
    except BaseException:
        if reducer is not None:
            reducer._armed = False       # no collective may be issued by a later, unrelated backward
        raise
    if reducer is not None:
        reducer.finalize()
    optimizer.step()
    _fused_mlp.invalidate_packed()       # folded Conv+BN weights cached by the fused inference kernels are stale now
    return loss.detach()


def synthetic_batch(n_frames, n_pts, dev, seed_base=0, n_obj=None):
    """Training batch with the tensor contract of the reference's dataset items (linemod_dataset.py:
    cld_rgb_nrm (N, 9), kp_targ_ofst (N, K, 3), ctr_targ_ofst (N, 1, 3), labels (N,))."""
    import numpy as np
    from . import synth
    n_obj = n_obj if n_obj is not None else n_pts // 4
    fr = [synth.synth_frame(frame=seed_base + i, n_pts=n_pts, n_obj=n_obj) for i in range(n_frames)]
    pc = np.stack([np.concatenate([f["pcld"], f["feats"].T], 1) for f in fr], 0).astype(np.float32)
    kp_t = np.stack([np.transpose(f["pred_kp_of"], (1, 0, 2)) for f in fr], 0).astype(np.float32)      # truth + noise
    ctr_t = np.stack([np.transpose(f["ctr_of"], (1, 0, 2)) for f in fr], 0).astype(np.float32)
    lab = np.stack([f["mask"] for f in fr], 0).astype(np.float32)[..., None]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return dict(pc=T(pc), kp_targ_ofst=T(kp_t), ctr_targ_ofst=T(ctr_t), labels=T(lab))


def algorithmic_work_per_step(backbone, n_frames, n_pts=12288):
    """Algorithmic FLOPs and bytes of ONE training step of the SA / FP SharedMLP chains (BASELINE config 5), i.e. of
    the part of the step this package owns (csrc/mlp_train.hip); the heads, the loss and the optimizer are not in it.

    FLOPs: per layer 2 * rows * cin * cout for the forward product, the same for the weight gradient, and the same for
    the input gradient wherever one is needed (every layer but the first of SA level 0, whose input -- the cloud's own
    features and coordinates -- takes no gradient).
    Bytes: what the bf16 formulation has to stream per layer (true channel counts, no padding; bf16 = 2 B):
      forward   GEMM reads X, writes Y (pre-BatchNorm: batch statistics need the whole matrix before any element can be
                normalised); BatchNorm+ReLU reads Y, writes H -- the pooled last layer of an SA chain reads Y and writes
                the pooled fp32 row + 1-byte arg-indices instead;
      backward  statistics pass reads dH and Y (pooled: Y at the arg positions, 1/nsample of it); apply pass reads dH
                and Y, writes dY (pooled: reads Y, writes dY); input gradient reads dY, writes dX; weight gradient
                reads dY and H_prev;
      layer 0   the gathered input X0 is written once (forward) and its gradient read once (backward), the fp32
                source / gradient rows once each.
    Weights and statistics are KBs and left out.  Returns dict(flops, bytes, per_level=[...]), rows = matrix rows."""
    scale = n_pts / 12288.0
    B = n_frames
    flops = bytes_ = 0.0
    levels = []

    def chain(name, rows, dims, pooled_ns, input_grad, src_rows, src_ch):
        nonlocal flops, bytes_
        f = b = 0.0
        L = len(dims) - 1
        b += rows * dims[0] * 2 + src_rows * src_ch * 4               # X0 written, fp32 source rows read
        for li in range(L):
            cin, cout = dims[li], dims[li + 1]
            last_pooled = pooled_ns and li == L - 1
            f += 2.0 * rows * cin * cout * (2 + (1 if (li > 0 or input_grad) else 0))
            b += rows * (cin + cout) * 2                               # forward GEMM
            if last_pooled:
                b += rows * cout * 2 + (rows // pooled_ns) * cout * 5  # Y read, pooled fp32 + arg written
                b += (rows // pooled_ns) * cout * (2 + 4 + 1)          # bwd statistics: Y at arg, dout, arg
                b += rows * cout * 2 * 2                               # bwd apply: Y read, dY written
            else:
                b += rows * cout * 2 * 2                               # BatchNorm + ReLU: Y read, H written
                b += rows * cout * 2 * 2                               # bwd statistics: dH, Y
                b += rows * cout * 2 * 3                               # bwd apply: dH, Y read, dY written
            if li > 0 or input_grad:
                b += rows * (cout + cin) * 2                           # input gradient: dY read, dX written
            b += rows * (cout + cin) * 2                               # weight gradient: dY, H_prev read
        if input_grad:
            b += rows * dims[0] * 2 + src_rows * src_ch * 4           # dX0 read, fp32 gradient rows written
        flops += f
        bytes_ += b
        levels.append(dict(name=name, rows=int(rows), dims=list(dims), flops=f, bytes=b))

    n_in = n_pts
    sa_width = []
    for li, mod in enumerate(backbone.SA_modules):
        m = int(mod.npoint * scale)
        for si, (grouper, mlp) in enumerate(zip(mod.groupers, mod.mlps)):
            convs = [c for c in mlp.modules() if isinstance(c, nn.Conv2d)]
            dims = [convs[0].in_channels] + [c.out_channels for c in convs]
            chain("SA%d.%d" % (li, si), B * m * grouper.nsample, dims, grouper.nsample, li > 0, B * n_in, dims[0] - 3)
            if si == 0:
                sa_width.append(0)
                if li == 0:
                    c_input = dims[0] - 3
            sa_width[-1] += dims[-1]
        n_in = m
    n_unknown = [int(v * scale) for v in (12288, 2048, 1024, 512)]
    n_known = [int(v * scale) for v in (2048, 1024, 512, 128)]
    for fi, mod in enumerate(backbone.FP_modules):
        convs = [c for c in mod.mlp.modules() if isinstance(c, nn.Conv2d)]
        dims = [convs[0].in_channels] + [c.out_channels for c in convs]
        # sources: the known points' features (interpolated, C2 channels) + the unknown points' own skip features (C1)
        c1 = c_input if fi == 0 else sa_width[fi - 1]
        c2 = dims[0] - c1
        src_floats = B * (n_known[fi] * c2 + n_unknown[fi] * c1)
        chain("FP%d" % fi, B * n_unknown[fi], dims, 0, True, src_floats, 1)
    return dict(flops=flops, bytes=bytes_, per_level=levels)
