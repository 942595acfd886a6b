"""Device-side vote -> cluster -> pose engine: thin host glue over section 3 of the C ABI
(include/pvn3d_hip.h).  The reference-API mirrors (meanshift_pytorch.MeanShiftTorch,
pvn3d_eval_utils.cal_frame_poses / cal_frame_poses_lm) call into this module.

Everything stays on the GPU: vote assembly and mask compaction (pvn3d_vote_compact), all
(K+1) mean-shift fits of every object of every frame in ONE batched call
(pvn3d_meanshift_fit_batch), and the Kabsch fit (pvn3d_best_fit_transform).  The only host
synchronisation is the optional convergence poll and the final read-back of the poses.
"""
import os
import threading

import numpy as np
import torch

from ..._lib import lib, check, on_device

_tls = threading.local()


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _poll_buf():
    """Per-thread pinned int[2] for the convergence poll (the reference's post-processing is
    called from a thread pool, pvn3d_eval_utils.py:373-380, so nothing here is shared)."""
    b = getattr(_tls, "poll", None)
    if b is None:
        b = torch.zeros(2, dtype=torch.int32).pin_memory()
        _tls.poll = b
    return b


MS_ALIGNED32 = 1    # include/pvn3d_hip.h PVN3D_MS_ALIGNED32
MS_NO_EARLY_OUT = 2  # PVN3D_MS_NO_EARLY_OUT: iterate every seed in every iteration
MS_FORCE_SCALAR = 4  # PVN3D_MS_FORCE_SCALAR: one seed per lane
MS_FORCE_PACKED = 8  # PVN3D_MS_FORCE_PACKED: two seeds per lane (packed fp32 math)
MS_FORCE_WHOLE = 16  # PVN3D_MS_FORCE_WHOLE: every wave walks all points of the fit
MS_FORCE_SPLIT = 32  # PVN3D_MS_FORCE_SPLIT: the four waves of a workgroup split the points
MS_SGPR_POINTS = 64  # PVN3D_MS_SGPR_POINTS: LDS-free iteration kernel (points as scalar operands); needs ALIGNED32
MS_NO_WINNER_STOP = 128  # PVN3D_MS_NO_WINNER_STOP: run the reference's full iteration count (iters == its `it`)
MS_COUNT_TWO_PASS = 1 << 28  # PVN3D_MS_COUNT_TWO_PASS: the neighbour count as the two launches of rounds 2-5 (cross-check)
# Iteration-kernel choice used when a call does not name one (None: the library default).  A pipelined evaluator
# that runs the vote stage beside the fused-MLP kernels sets "sgpr+cap<waves>" here (bench.py does).
DEFAULT_KERNEL = None
SGPR_MIN_FITS = 128        # a default of "sgpr" is only followed for batches of more fits than this


def meanshift_fit_batch(pts4, seg_off, seg_cnt, max_cnt, bandwidth, max_iter=300, labels=None,
                        poll_every=8, aligned32=False, kernel=None, enqueue_limit=None):
    """Batched MeanShiftTorch.fit.

    pts4 (total,4) float32 cuda; seg_off/seg_cnt (n_seg) int32 cuda; max_cnt: host bound on
    seg_cnt.  Returns ctr (n_seg,3) float32, labels (total) uint8, iters (n_seg) int32.
    poll_every = 0 -> fully asynchronous (enqueues max_iter+1 iterations).
    enqueue_limit = E -> no host poll and at most E iterations enqueued (a fixed launch sequence: capturable in a HIP
    graph); fits that would still run come back with a NEGATIVE iteration count and a centre that is not final.
    aligned32: every seg_off is a multiple of 32 and each segment owns roundup32(cnt) rows.
    kernel: None (library default) or a '+'-joined choice of "scalar" | "packed", "whole" | "split" and
    "noearly" (no early-out of converged seeds) -- pins the iteration kernel variant (all give identical results);
    "nowin" = no winner stop: iterate until the reference's stop rule says so (iters == the reference's `it`; same
    centres and labels, bit for bit);
    "sgpr" = the LDS-free kernel (needs aligned32), "cap<n>" bounds its launch to n waves (PVN3D_MS_WAVE_CAP);
    "count2" = the neighbour count of the original points as two launches (rounds 2-5) instead of the symmetric one.
    """
    flags = MS_ALIGNED32 if aligned32 else 0
    if kernel is None:
        kernel = DEFAULT_KERNEL
        # the LDS-free default only applies where the layout allows it, and where there are fits enough to fill the chip
        # with one wave per 128 seeds: below ~128 fits the LDS kernel, whose four waves per tile split the points, is the
        # shorter launch (tools/ms_rate.py: 64 fits of 3072 votes 0.131 vs 0.173 ms per iteration, 288 fits 0.530 vs 0.519)
        if kernel is not None and "sgpr" in kernel and (not aligned32 or int(seg_off.numel()) <= SGPR_MIN_FITS):
            rest = [k for k in kernel.split("+") if k != "sgpr" and not k.startswith("cap")]
            kernel = "+".join(rest) if rest else None
    if kernel is not None:
        for k in kernel.split("+"):
            if k.startswith("cap"):
                flags |= (int(k[3:]) & 0xfffff) << 8
                continue
            flags |= {"scalar": MS_FORCE_SCALAR, "packed": MS_FORCE_PACKED, "whole": MS_FORCE_WHOLE,
                      "split": MS_FORCE_SPLIT, "noearly": MS_NO_EARLY_OUT, "sgpr": MS_SGPR_POINTS,
                      "nowin": MS_NO_WINNER_STOP, "count2": MS_COUNT_TWO_PASS}[k]
    dev = pts4.device
    assert pts4.is_cuda and pts4.dtype == torch.float32 and pts4.is_contiguous() and pts4.size(1) == 4
    n_seg = int(seg_off.numel())
    total = int(pts4.size(0))
    ctr = torch.empty((n_seg, 3), dtype=torch.float32, device=dev)
    iters = torch.empty((n_seg,), dtype=torch.int32, device=dev)
    if labels is None:
        labels = torch.empty((total,), dtype=torch.uint8, device=dev)
    if n_seg == 0:
        return ctr, labels, iters
    ws_bytes = int(lib.pvn3d_meanshift_workspace_bytes(n_seg, total, int(max_iter)))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    poll = _poll_buf() if poll_every > 0 and enqueue_limit is None else None
    if enqueue_limit is not None:
        poll_every = int(enqueue_limit)
    with on_device(dev):
        check(lib.pvn3d_meanshift_fit_batch(
            pts4.data_ptr(), seg_off.data_ptr(), seg_cnt.data_ptr(), n_seg, total, int(max_cnt),
            float(bandwidth), int(max_iter), ctr.data_ptr(), labels.data_ptr(), iters.data_ptr(),
            ws.data_ptr(), ws_bytes, poll.data_ptr() if poll is not None else None,
            int(poll_every), int(flags), _stream(dev)), "meanshift_fit_batch")
    return ctr, labels, iters


def seg_stride_rows(n_pts):
    """Rows a vote segment occupies: n_pts rounded up to 32 (the PVN3D_MS_ALIGNED32 promise), plus 32 when that is a
    multiple of 1024 rows (16 KiB): segments that start a multiple of 2^14 .. 2^16 bytes apart make the iteration
    kernels' waves -- one fit each, walking their points at the same pace -- hit the same memory channels at the same time
    (measured on the headline batch, tools/ms_rate.py: 1.26 ms per iteration at 12288 rows per segment, 1.00 ms at 12320)."""
    s = (int(n_pts) + 31) // 32 * 32
    return s + 32 if s % 1024 == 0 else s


def vote_compact(pcld, mask, ctr_of, pred_kp_of, inst_frame, inst_cls, v_first, v_count,
                 sel=None, sel_inst_stride=0, out=None):
    """pvn3d_vote_compact_strided wrapper.  Returns (votes (n_seg*stride,4), seg_off, seg_cnt) where
    n_seg = n_inst*(n_kps+1) and stride = seg_stride_rows(n_pts); only segments v_first..v_first+v_count-1 of each
    instance are (re)written.  sel / sel_inst_stride: labels of an earlier fit batch on this layout
    (sel_inst_stride = (n_kps + 1) * stride)."""
    dev = pcld.device
    F, n_pts = pcld.size(0), pcld.size(1)
    n_kps = pred_kp_of.size(1)
    n_inst = int(inst_frame.numel())
    n_seg = n_inst * (n_kps + 1)
    stride = seg_stride_rows(n_pts)
    if out is None:
        votes = torch.empty((n_seg * stride, 4), dtype=torch.float32, device=dev)
        seg = torch.zeros((2, n_seg), dtype=torch.int32, device=dev)      # one fill; rows = offsets, counts
        seg_off, seg_cnt = seg[0], seg[1]
    else:
        votes, seg_off, seg_cnt = out
    with on_device(dev):
        check(lib.pvn3d_vote_compact_strided(
            F, n_pts, stride, n_kps, n_inst, int(v_first), int(v_count), pcld.data_ptr(), mask.data_ptr(),
            ctr_of.data_ptr(), pred_kp_of.data_ptr(), inst_frame.data_ptr(), inst_cls.data_ptr(),
            sel.data_ptr() if sel is not None else None, int(sel_inst_stride), votes.data_ptr(),
            seg_off.data_ptr(), seg_cnt.data_ptr(), _stream(dev)), "vote_compact")
    return votes, seg_off, seg_cnt


def best_fit_transform_batch(A, B, valid=None):
    """A, B (S,npts,3) float32 cuda -> T (S,3,4) float64 cuda; valid (S) int32 or None."""
    dev = A.device
    S, npts = A.size(0), A.size(1)
    T = torch.empty((S, 3, 4), dtype=torch.float64, device=dev)
    with on_device(dev):
        check(lib.pvn3d_best_fit_transform(S, npts, A.data_ptr(), B.data_ptr(),
                                           valid.data_ptr() if valid is not None else None,
                                           T.data_ptr(), _stream(dev)), "best_fit_transform")
    return T


_const_cache = {}


def _device_const(key, dev, make):
    """Small constant tensors (mesh keypoints, thresholds, index ramps) uploaded once per device: a host-to-device copy
    per call would also make the single-frame path uncapturable."""
    k = (key, str(dev))
    t = _const_cache.get(k)
    if t is None:
        t = _const_cache[k] = make().to(dev)
    return t


def _prep(pcld, mask, ctr_of, pred_kp_of):
    pcld = pcld.contiguous().float()
    mask = mask.contiguous().to(torch.int32)
    ctr_of = ctr_of.contiguous().float()
    pred_kp_of = pred_kp_of.contiguous().float()
    return pcld, mask, ctr_of, pred_kp_of


def frames_pose_single_class(pcld, mask, ctr_of, pred_kp_of, mesh_kps, cls_id=1, use_ctr=True,
                             use_ctr_clus_flter=False, radius=0.08, max_iter=300, poll_every=8, async_limit=None):
    """Batched cal_frame_poses_lm (pvn3d/lib/utils/pvn3d_eval_utils.py:156-201).

    pcld (F,N,3); mask (F,N) integer; ctr_of (F,1,N,3); pred_kp_of (F,K,N,3);
    mesh_kps (K+use_ctr,3) object-frame keypoints (+centre last).
    Returns dict(poses (F,3,4) f64 cuda, cls_kps (F,K+1,3), iters (F,K+1), counts (F,K+1), unfinished_min 0-dim).
    async_limit = E: no host poll, at most E iterations per fit batch enqueued; `iters` < 0 marks fits that did not
    finish (their poses are not final; `unfinished_min` = iters.min() < 0 then) -- the launch sequence is then fixed
    and capturable (GraphedFramePoses).
    """
    pcld, mask, ctr_of, pred_kp_of = _prep(pcld, mask, ctr_of, pred_kp_of)
    dev = pcld.device
    F, N = pcld.size(0), pcld.size(1)
    K = pred_kp_of.size(1)
    inst_frame = _device_const(("ramp", F), dev, lambda: torch.arange(F, dtype=torch.int32))
    inst_cls = _device_const(("cls", F, int(cls_id)), dev, lambda: torch.full((F,), int(cls_id), dtype=torch.int32))
    if not use_ctr_clus_flter:
        votes, seg_off, seg_cnt = vote_compact(pcld, mask, ctr_of, pred_kp_of, inst_frame,
                                               inst_cls, 0, K + 1)
        ctr, _, iters = meanshift_fit_batch(votes, seg_off, seg_cnt, N, radius, max_iter,
                                            poll_every=poll_every, aligned32=True, enqueue_limit=async_limit)
    else:
        out = vote_compact(pcld, mask, ctr_of, pred_kp_of, inst_frame, inst_cls, K, 1)
        votes, seg_off, seg_cnt = out
        so = seg_off.view(F, K + 1)
        sc = seg_cnt.view(F, K + 1)
        c_ctr, labels, it_ctr = meanshift_fit_batch(votes, so[:, K].contiguous(),
                                                    sc[:, K].contiguous(), N, radius, max_iter,
                                                    poll_every=poll_every, aligned32=True,
                                                    enqueue_limit=async_limit)
        # keypoint votes filtered by the centre fit's inlier labels (rows of segment K)
        sel = labels[K * seg_stride_rows(N):]
        vote_compact(pcld, mask, ctr_of, pred_kp_of, inst_frame, inst_cls, 0, K, sel=sel,
                     sel_inst_stride=(K + 1) * seg_stride_rows(N), out=out)
        c_kp, _, it_kp = meanshift_fit_batch(votes, so[:, :K].contiguous().view(-1),
                                             sc[:, :K].contiguous().view(-1), N, radius, max_iter,
                                             poll_every=poll_every, aligned32=True,
                                             enqueue_limit=async_limit)
        ctr = torch.cat([c_kp.view(F, K, 3), c_ctr.view(F, 1, 3)], 1).view(-1, 3)
        iters = torch.cat([it_kp.view(F, K), it_ctr.view(F, 1)], 1).view(-1)
    cls_kps = ctr.view(F, K + 1, 3)
    counts = seg_cnt.view(F, K + 1)
    valid = (mask == int(cls_id)).any(dim=1).to(torch.int32)
    npts = K + 1 if use_ctr else K
    mesh_dev = mesh_kps if mesh_kps.device == dev else _device_const(
        ("mesh1", mesh_kps.detach().to(torch.float32).numpy().tobytes()), dev, lambda: mesh_kps.to(torch.float32))
    A = mesh_dev.to(torch.float32)[:npts].unsqueeze(0).expand(F, npts, 3).contiguous()
    B = cls_kps[:, :npts].contiguous()
    poses = best_fit_transform_batch(A, B, valid)
    return dict(poses=poses, cls_kps=cls_kps, iters=iters.view(F, K + 1), counts=counts, unfinished_min=iters.min())


def relabel_by_centre(pcld, ctr_of0, mask, ctrs, present, thr_lst):
    """Centre-cluster re-labelling of cal_frame_poses (pvn3d_eval_utils.py:58-72), one launch for
    every frame (csrc/relabel.hip); every class slot stays on the device.
    pcld, ctr_of0 (F,N,3); mask (F,N) int32; ctrs (F,C,3) cluster centre per class id 1..C;
    present (F,C) bool; thr_lst (C) float32 = fp32(0.8 * ycb_r_lst).
    Returns (new mask (F,N) int32, present_new (F,C) bool)."""
    F, N = mask.shape
    C = ctrs.size(1)
    new_mask = torch.empty_like(mask)
    present_new = torch.empty((F, C), dtype=torch.int32, device=mask.device)
    with on_device(mask.device):
        check(lib.pvn3d_relabel_by_centre(F, N, C, pcld.data_ptr(), ctr_of0.contiguous().data_ptr(), mask.data_ptr(),
                                          ctrs.contiguous().data_ptr(), present.to(torch.int32).contiguous().data_ptr(),
                                          thr_lst.contiguous().data_ptr(), new_mask.data_ptr(),
                                          present_new.data_ptr(), _stream(mask.device)), "relabel_by_centre")
    return new_mask, present_new != 0


def frames_pose_multi_class(pcld, mask, ctr_of, pred_kp_of, mesh_kps_all, n_cls, radius_lst,
                            use_ctr=True, use_ctr_clus_flter=True, radius=0.08, max_iter=300,
                            poll_every=8, async_limit=None):
    """Batched cal_frame_poses (pvn3d/lib/utils/pvn3d_eval_utils.py:37-110) for every class id
    1..n_cls-1 of every frame; absent classes are empty segments.

    mesh_kps_all (n_cls-1, K+1, 3): object-frame keypoints (+centre) per class id.
    Returns dict(poses (F,n_cls-1,3,4), present (F,n_cls-1) bool [original mask], cls_kps (F,n_cls-1,K+1,3),
                 iters, unfinished_min, new_mask).
    """
    pcld, mask, ctr_of, pred_kp_of = _prep(pcld, mask, ctr_of, pred_kp_of)
    dev = pcld.device
    F, N = pcld.size(0), pcld.size(1)
    K = pred_kp_of.size(1)
    C = n_cls - 1
    cls_ids = _device_const(("cls_ids", C), dev, lambda: torch.arange(1, n_cls, dtype=torch.int32)).view(1, C, 1)
    present = (mask.unsqueeze(1) == cls_ids).any(dim=2)                        # (F,C)
    present0 = present          # pred_cls_ids of the reference come from the ORIGINAL mask (:49)
    # Only the (frame, class) pairs that occur get an instance slot: ONE small device->host copy per batch
    # (F*C flags; the reference synchronises per class and per mean-shift iteration).  With every class
    # slot instantiated the vote buffer alone is F*C*(K+1)*N*16 bytes (2.4 GB at 64 frames, 21 classes)
    # and every launch carries 21 - (objects in view) empty segments per frame.
    # A single frame (the reference's test_mini_batch_size = 1) instantiates all C class slots instead (40 MB of
    # votes): no host round trip at all, the absent classes are empty segments, and -- the instance list being the
    # class list -- none of the scatter / gather steps below (each one a launch, ~5 us of a ~0.8 ms call).
    single = F == 1
    if single:
        inst_frame = _device_const(("inst_frame0", C), dev, lambda: torch.zeros(C, dtype=torch.int32))
        inst_cls = _device_const(("inst_cls", C), dev, lambda: torch.arange(1, C + 1, dtype=torch.int32))
        pf = pc = None
        n_inst = C
    else:
        pairs = torch.nonzero(present.cpu())                                    # (n_inst, 2) on the host
        n_inst = int(pairs.size(0))
        if n_inst == 0:
            poses_full = torch.zeros((F, C, 3, 4), dtype=torch.float64, device=dev)
            poses_full[:, :, 0, 0] = 1.0
            poses_full[:, :, 1, 1] = 1.0
            poses_full[:, :, 2, 2] = 1.0
            zi = torch.zeros((F, C, K + 1), dtype=torch.int32, device=dev)
            return dict(poses=poses_full, present=present0, present_new=present,
                        cls_kps=torch.zeros((F, C, K + 1, 3), dtype=torch.float32, device=dev), iters=zi,
                        unfinished_min=zi.min(), new_mask=mask)
        pf = pairs[:, 0].to(device=dev, dtype=torch.long)
        pc = pairs[:, 1].to(device=dev, dtype=torch.long)
        inst_frame = pf.to(torch.int32)
        inst_cls = (pc + 1).to(torch.int32)
    # a scalar-lane iteration kernel for the two centre batches of a single frame: at most N seeds in all, which
    # leaves most SIMDs idle either way, and one seed per lane is the shorter dependent chain (identical bits)
    ctr_kernel = "scalar" if single and DEFAULT_KERNEL is None else None

    n_seg = n_inst * (K + 1)
    votes = torch.empty((n_seg * seg_stride_rows(N), 4), dtype=torch.float32, device=dev)
    seg = torch.zeros((2, n_seg), dtype=torch.int32, device=dev)          # segment offsets, counts
    out = (votes, seg[0], seg[1])

    def seg_slices(lo, hi):
        """(offsets, counts) of segments lo..hi-1 of every instance: one gather for both tables"""
        x = seg.view(2, n_inst, K + 1)[:, :, lo:hi].contiguous().view(2, -1)
        return x[0], x[1]

    it0 = None
    if use_ctr_clus_flter:
        vote_compact(pcld, mask, ctr_of, pred_kp_of, inst_frame, inst_cls, K, 1, out=out)
        o_c, n_c = seg_slices(K, K + 1)
        c0, _, it0 = meanshift_fit_batch(votes, o_c, n_c, N, radius, max_iter, poll_every=poll_every,
                                         aligned32=True, enqueue_limit=async_limit, kernel=ctr_kernel)
        thr = _device_const(("ycb_thr", tuple(np.asarray(radius_lst, np.float64).tolist())), dev,
                            lambda: torch.from_numpy((np.asarray(radius_lst, np.float64) * 0.8).astype(np.float32)))
        if single:
            ctrs = c0.view(1, C, 3)
        else:
            ctrs = torch.zeros((F, C, 3), dtype=torch.float32, device=dev)
            ctrs[pf, pc] = c0
        mask, present = relabel_by_centre(pcld, ctr_of[:, 0], mask, ctrs, present, thr)
    # per-class centre fit on the (re-labelled) mask
    vote_compact(pcld, mask, ctr_of, pred_kp_of, inst_frame, inst_cls, K, 1, out=out)
    o_c, n_c = seg_slices(K, K + 1)
    c_ctr, labels, it_ctr = meanshift_fit_batch(votes, o_c, n_c, N, radius, max_iter, poll_every=poll_every,
                                                aligned32=True, enqueue_limit=async_limit, kernel=ctr_kernel)
    sel = labels[K * seg_stride_rows(N):] if use_ctr_clus_flter else None
    vote_compact(pcld, mask, ctr_of, pred_kp_of, inst_frame, inst_cls, 0, K, sel=sel,
                 sel_inst_stride=(K + 1) * seg_stride_rows(N), out=out)
    o_k, n_k = seg_slices(0, K)
    c_kp, _, it_kp = meanshift_fit_batch(votes, o_k, n_k, N, radius, max_iter,
                                         poll_every=poll_every, aligned32=True, enqueue_limit=async_limit)
    cls_kps = torch.cat([c_kp.view(n_inst, K, 3), c_ctr.view(n_inst, 1, 3)], 1)
    # iteration counts of every fit batch of the call in one table: columns 0..K are `iters`; the filter pass (whose
    # centres only re-label the mask) is the last column.  Under async_limit a negative count marks a fit that did not
    # finish; `unfinished_min` < 0 says that some fit of the call -- the filter pass included, which taints everything
    # after it -- did not, i.e. the call's results are not final.
    cols = [it_kp.view(n_inst, K), it_ctr.view(n_inst, 1)] + ([it0.view(n_inst, 1)] if it0 is not None else [])
    all_it = torch.cat(cols, 1)
    iters = all_it[:, :K + 1]
    unfinished_min = all_it.min()
    npts = K + 1 if use_ctr else K
    mesh_dev = mesh_kps_all if mesh_kps_all.device == dev else _device_const(
        ("mesh_all", mesh_kps_all.detach().to(torch.float32).numpy().tobytes()), dev, lambda: mesh_kps_all.to(torch.float32))
    mesh_dev = mesh_dev.to(torch.float32)
    A = (mesh_dev[:, :npts] if single else mesh_dev[pc, :npts]).contiguous()
    B = cls_kps[:, :npts].contiguous()
    valid = (present[0] if single else present[pf, pc]).to(torch.int32).contiguous()
    poses = best_fit_transform_batch(A, B, valid)
    if single:
        poses_full, kps_full, iters_full = poses.unsqueeze(0), cls_kps.unsqueeze(0), iters.unsqueeze(0)
    else:
        poses_full = torch.zeros((F, C, 3, 4), dtype=torch.float64, device=dev)
        poses_full[:, :, 0, 0] = 1.0
        poses_full[:, :, 1, 1] = 1.0
        poses_full[:, :, 2, 2] = 1.0
        kps_full = torch.zeros((F, C, K + 1, 3), dtype=torch.float32, device=dev)
        iters_full = torch.zeros((F, C, K + 1), dtype=torch.int32, device=dev)
        poses_full[pf, pc] = poses
        kps_full[pf, pc] = cls_kps
        iters_full[pf, pc] = iters.to(torch.int32)
    return dict(poses=poses_full, present=present0, present_new=present, cls_kps=kps_full, iters=iters_full,
                unfinished_min=unfinished_min, new_mask=mask)


def add_adds_batch(pts_list, pred_RT, gt_RT):
    """ADD / ADD-S of a batch of instances in one launch (csrc/metrics.hip).
    pts_list: list of (n_i,3) float32 CUDA tensors (mesh points per instance) or one (n,3) tensor
    shared by all; pred_RT / gt_RT: (n_inst,3,4) float32 CUDA.  Returns (add, adds) (n_inst,) float32
    CUDA tensors; no host sync."""
    pred_RT = pred_RT.to(torch.float32).contiguous()
    gt_RT = gt_RT.to(torch.float32).contiguous()
    n_inst = pred_RT.size(0)
    dev = pred_RT.device
    if torch.is_tensor(pts_list):
        pts_list = [pts_list] * n_inst
    counts = [int(p.size(0)) for p in pts_list]
    pts = torch.cat([p.to(torch.float32).reshape(-1, 3) for p in pts_list], 0).contiguous()
    off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0).tolist()), dtype=torch.int32, device=dev)
    max_pts = max(counts) if counts else 0
    add = torch.zeros(n_inst, dtype=torch.float32, device=dev)
    adds = torch.zeros(n_inst, dtype=torch.float32, device=dev)
    if n_inst == 0 or max_pts == 0:
        return add, adds
    wsb = lib.pvn3d_add_adds_workspace_bytes(n_inst, max_pts)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with on_device(dev):
        check(lib.pvn3d_add_adds_batch(n_inst, max_pts, pts.data_ptr(), off.data_ptr(), pred_RT.data_ptr(),
                                       gt_RT.data_ptr(), ws.data_ptr(), wsb, add.data_ptr(), adds.data_ptr(),
                                       _stream(dev)), "add_adds_batch")
    return add, adds
