#!/usr/bin/env python3
"""bench.py -- frames/s of the PVN3D per-point voting hot path on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (for N>1 launched by
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...``, one rank per
GPU).  W untimed steps, then EXACTLY K timed steps bracketed by barrier + synchronize, MAX over
ranks, rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1], "LineMOD 'ape' eval path, N=12 288 pts"): one STEP = one
batch of ``--frames`` synthetic frames per GPU, resident in HBM, through the hot path:
  (A) the Pointnet2MSG forward of lib/pvn3d.py:46-154 (random-init weights, eval mode) -- 4
      multi-scale set-abstraction levels (FPS -> gather -> two-radius ball_query -> fused
      [group xyz+features -> SharedMLP -> max-pool] on fp32 MFMA) and 4 feature-propagation
      levels (three_nn -> weights -> fused [three_interpolate -> concat -> SharedMLP]);
      ``--ops-only`` replaces it by the bare op chain with synthetic feature tensors (the
      round-1 first measurement, no GEMMs);
  (B) vote assembly -> (K+1) MeanShift fits per frame -> Kabsch pose (cal_frame_poses_lm).
The HBM rooflines of the data-movement ops (ball_query, group, three_interpolate, ...) are
measured on the unfused op chain in a separate, untimed-for-`value` pass, because in (A)
grouping and interpolation never touch HBM (they feed the MFMA kernel through LDS).
Frames are independent, so N GPUs run N x frames per step with no data-path collective (weak
scaling); the only communication is the timing reduction.

``--strong`` keeps the TOTAL number of frames per step at ``--frames`` (BASELINE config 4: 64 frames per
batch sharded over the ranks) instead of ``--frames`` per GPU; at N > 1 every step ends with the per-frame
result all-gather (pvn3d_amd/sharding.py) inside the timed region -- the only collective of the path.

Extra JSON objects: ``roofline`` (dominant kernel, measured with events inside the timed region),
``rooflines`` (every stage), ``cpu_baseline`` (the reference's dense-torch MeanShift + numpy
Kabsch restated in oracle/, timed on the host cores on a bounded sample; rank 0, N=1 only),
``configs`` (rank 0, N=1: the other BASELINE configurations, each on a few steps: B=1 latency, the
n_obj = 12 288 stress case, the YCB 5-object frame, config 1 (N = 2048) with its CPU time in full, a
heavy-tailed vote set with its measured MeanShift iteration counts) and ``device_copy`` (measured
HBM copy / fill bandwidth on this device, the achievable roofline next to the 8 TB/s peak).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
PEAK_FP32_VALU_TFLOPS = 157.3
PEAK_FP32_MFMA_TFLOPS = 157.3      # v_mfma_f32_32x32x2_f32, exact fp32 (guide: matrix fp32 = vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA (guide; the 2:1-sparsity headline figure is not used)
# VALU issue bound of the pair-scanning ops ("report time and % of a VALU bound, not GB/s"):
# one fp32 VALU instruction per lane and clock on each of 256 CUs x 4 SIMDs x 32 lanes at 2.4 GHz (= the 157.3
# TFLOP/s peak counted as FMAs), and the fewest instructions the reference arithmetic allows per (query, point)
# pair: 3 sub + 3 mul + 2 add for the unfused squared distance, + 1 compare / min.
VALU_LANE_INSTR_PER_S = 256 * 4 * 32 * 2.4e9
VALU_INSTR_PER_PAIR = {"fps": 9, "three_nn": 9, "ball_query": 9}


def pair_evals_per_frame(n_pts):
    """(query, point) distance evaluations of the reference's brute-force kernels per frame (SURVEY.md 8d:
    27.8 M / 27.9 M / 55.7 M at N = 12288)."""
    s = n_pts / 12288.0
    sa = [(int(12288 * s), int(2048 * s)), (int(2048 * s), int(1024 * s)), (int(1024 * s), int(512 * s)),
          (int(512 * s), int(128 * s))]
    fps = sum(n * (m - 1) for n, m in sa)
    bq = 2 * sum(n * m for n, m in sa)          # two radii per level
    nn = sum(n * m for n, m in [(int(512 * s), int(128 * s)), (int(1024 * s), int(512 * s)),
                                (int(2048 * s), int(1024 * s)), (int(12288 * s), int(2048 * s))])
    return {"fps": fps, "ball_query": bq, "three_nn": nn}

# Pointnet2MSG hyper-parameters, pvn3d/lib/pvn3d.py:65-118 (input_channels = 6)
SA_LEVELS = [  # (n_in, npoint, C_in, radii, nsamples, C_out)
    (12288, 2048, 6, (0.0175, 0.025), (16, 32), 96),
    (2048, 1024, 96, (0.025, 0.05), (16, 32), 256),
    (1024, 512, 256, (0.05, 0.1), (16, 32), 512),
    (512, 128, 512, (0.1, 0.2), (16, 32), 1024),
]
FP_LEVELS = [  # (n unknown, m known, C_known) in execution order (pvn3d.py:149-152)
    (512, 128, 1024), (1024, 512, 512), (2048, 1024, 512), (12288, 2048, 256),
]


def algorithmic_bytes_per_frame(n_pts):
    """SURVEY.md section 8(d) formulas, fp32 / int32, scaled to n_pts input points."""
    s = n_pts / 12288.0
    out = dict(ball_query=0.0, group=0.0, fps=0.0, gather=0.0, three_nn=0.0, three_interpolate=0.0)
    for (n_in, m, c, _r, nss, _co) in SA_LEVELS:
        n_in, m = int(n_in * s), int(m * s)
        out["fps"] += 12 * n_in + 4 * m
        out["gather"] += 4 * m + 24 * m
        for ns in nss:
            out["ball_query"] += 12 * m + 12 * n_in + 4 * m * ns
            out["group"] += (4 * m * ns + 4 * 3 * n_in + 4 * 3 * m * ns) + (4 * m * ns + 4 * c * n_in + 4 * c * m * ns)
    for (n, m, c) in FP_LEVELS:
        n, m = int(n * s), int(m * s)
        out["three_nn"] += 12 * n + 12 * m + 24 * n
        out["three_interpolate"] += 4 * c * m + 24 * n + 4 * c * n
    return out


class StageTimer(object):
    """Event pairs on torch's current stream (the stream every kernel here is launched on)."""

    def __init__(self, enabled):
        self.enabled = enabled
        self.pairs = {}

    def start(self, name):
        if not self.enabled:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self.pairs.setdefault(name, []).append((e0, e1))
        return e1

    @staticmethod
    def stop(e1):
        if e1 is not None:
            e1.record()

    def totals_ms(self):
        return {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self.pairs.items()}

    def counts(self):
        return {k: len(v) for k, v in self.pairs.items()}


def make_inputs(n_frames, n_pts, n_obj, dev, seed_base):
    from pvn3d_amd import synth
    frames = [synth.synth_frame(frame=seed_base + i, n_pts=n_pts, n_obj=n_obj) for i in range(n_frames)]
    st = lambda k, dt=None: torch.from_numpy(np.stack([f[k] for f in frames], 0)).to(dev)
    s = n_pts / 12288.0
    g = torch.Generator(device="cpu").manual_seed(1234 + seed_base)
    inp = dict(pcld=st("pcld").contiguous(), mask=st("mask").to(torch.int32).contiguous(),
               ctr_of=st("ctr_of").contiguous(), pred_kp_of=st("pred_kp_of").contiguous(),
               feats=st("feats").contiguous(), frames=frames)
    # synthetic per-level feature tensors with the network's widths (MLP outputs are "next")
    inp["sa_feats"] = [torch.randn((n_frames, co, int(m * s)), generator=g).to(dev)
                       for (_n, m, _c, _r, _ns, co) in SA_LEVELS]
    inp["fp_known"] = [torch.randn((n_frames, c, int(m * s)), generator=g).to(dev) for (_n, m, c) in FP_LEVELS]
    return inp


def run_ops(inp, timer, scale):
    """(A) the SA/FP operator chain.  Returns small checksums so nothing is dead."""
    from pvn3d_amd.lib.pointnet2_utils import _ext
    xyz = inp["pcld"]
    feats = inp["feats"]
    l_xyz = [xyz]
    keep = []
    for li, (n_in, m, c, radii, nss, _co) in enumerate(SA_LEVELS):
        m = int(m * scale)
        t = timer.start("fps")
        sel = _ext.furthest_point_sampling(xyz, m)
        timer.stop(t)
        t = timer.start("gather")
        new_xyz = _ext.gather_points(xyz.transpose(1, 2).contiguous(), sel).transpose(1, 2).contiguous()
        timer.stop(t)
        t = timer.start("ball_query")
        i0, i1 = _ext.ball_query_pair(new_xyz, xyz, radii[0], nss[0], radii[1], nss[1])
        timer.stop(t)
        t = timer.start("group")
        g0, g1 = _ext.group_xyz_features_pair(xyz, new_xyz, feats, i0, i1)      # both radii of the level, one launch
        timer.stop(t)
        keep.append((g0, g1))
        xyz = new_xyz
        feats = inp["sa_feats"][li]
        l_xyz.append(xyz)
    for fi, (n, mm, c) in enumerate(FP_LEVELS):
        unknown, known = l_xyz[3 - fi], l_xyz[4 - fi]
        t = timer.start("three_nn")
        d2, idx = _ext.three_nn(unknown, known)
        timer.stop(t)
        dist_recip = 1.0 / (torch.sqrt(d2) + 1e-8)      # pointnet2_utils.py:126, modules :184-186
        weight = (dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)).contiguous()
        t = timer.start("three_interpolate")
        up = _ext.three_interpolate(inp["fp_known"][fi], idx, weight)
        timer.stop(t)
        keep.append(up)
    return keep


class _HookStage(object):
    """Context manager handed to pointnet2_modules.STAGE_HOOK: one event pair per stage."""

    def __init__(self, timer, name):
        self.timer, self.name = timer, name

    def __enter__(self):
        self.e1 = self.timer.start(self.name)

    def __exit__(self, *exc):
        StageTimer.stop(self.e1)
        return False


def make_net(dev):
    """Pointnet2MSG with the reference's hyper-parameters, random-init weights (seeded), eval mode."""
    from pvn3d_amd.lib.pointnet2_msg import Pointnet2MSG
    torch.manual_seed(20260925)
    return Pointnet2MSG(input_channels=6).to(dev).eval()


def run_net(net, inp, timer, geometry=None):
    """(A) the fused Pointnet2MSG forward; `timer` (if enabled) brackets its stages.  geometry: a handle from
    net.geometry_ahead(inp["pc"]) -- this batch's xyz-only work, enqueued one step earlier."""
    from pvn3d_amd.lib.pointnet2_utils import pointnet2_modules as pm
    from pvn3d_amd.lib import pointnet2_msg
    pm.STAGE_HOOK = (lambda name: _HookStage(timer, name)) if timer.enabled else None
    ahead = pointnet2_msg.GEOMETRY_STREAM
    if timer.enabled:
        pointnet2_msg.GEOMETRY_STREAM = False     # stage events need one serial stream
    try:
        with torch.no_grad():
            t = timer.start("pointnet2_msg_total")
            out = net(inp["pc"], geometry=geometry)
            timer.stop(t)
    finally:
        pm.STAGE_HOOK = None
        pointnet2_msg.GEOMETRY_STREAM = ahead
    return out


def mlp_flops_per_frame(net, scale):
    """2*K*M flops per column of every 1x1-conv layer x the columns it runs on."""
    sa = fp = 0.0
    for mod in net.SA_modules:
        m = int(mod.npoint * scale)
        for grouper, mlp in zip(mod.groupers, mod.mlps):
            per_col = sum(2.0 * c.weight.shape[0] * c.weight.shape[1]
                          for c in mlp.modules() if isinstance(c, (torch.nn.Conv2d, torch.nn.Conv1d)))
            sa += per_col * m * grouper.nsample
    n_unknown = [int(v * scale) for v in (12288, 2048, 1024, 512)]    # FP_modules[0..3] outputs
    for mod, n in zip(net.FP_modules, n_unknown):
        per_col = sum(2.0 * c.weight.shape[0] * c.weight.shape[1]
                      for c in mod.mlp.modules() if isinstance(c, (torch.nn.Conv2d, torch.nn.Conv1d)))
        fp += per_col * n
    return sa, fp


def mlp_chain_table(net, scale, frames=64):
    """One row per fused chain of the forward: stage, widths, algorithmic flops per frame and the matrix pipe it runs on
    (asked of the library: pvn3d_mlp_split_ok / fp_layerwise_shape_ok are the dispatch tests of
    lib/pointnet2_utils/_ext.py; `frames` = frames per forward, part of the second test)."""
    from pvn3d_amd._lib import lib
    from pvn3d_amd.lib.pointnet2_utils import _fused_mlp
    import ctypes
    rows = []

    def add(stage, name, dims, cols, is_sa, c_a, c_b, ns):
        from pvn3d_amd.lib.pointnet2_utils import _ext
        arr = (ctypes.c_int * len(dims))(*dims)
        args = (1 if is_sa else 0, c_a, c_b, ns, len(dims) - 1, arr)
        split2 = (_fused_mlp.MLP_ARITH == "fp16x2" and (_ext.SPLIT2_NARROW or dims[1] >= 128)
                  and bool(lib.pvn3d_mlp_split2_ok(*args, _ext._mlp_flags())))
        split = not split2 and _fused_mlp.split_arith() and bool(lib.pvn3d_mlp_split_ok(*args))
        # FP levels 2-3: layer by layer on the split GEMM (csrc/split_gemm.hip).  flops_per_frame stays the
        # reference's formulation (conv over [interp; skip] on the unknown points); the launches execute fewer (the
        # first conv's interpolated half runs over the known points)
        layerwise = not split and not split2 and not is_sa and _ext.fp_layerwise_shape_ok(cols * frames, c_b, dims)
        fl = 2.0 * sum(a * b for a, b in zip(dims[:-1], dims[1:])) * cols
        if split2 or (layerwise and _fused_mlp.MLP_ARITH == "fp16x2"):
            kind = ("fp16x2 split on v_mfma_f32_32x32x16_f16 (3 partial products per multiply)"
                    + (", layer by layer" if layerwise else ""))
            peak = PEAK_BF16_MFMA_TFLOPS / 3.0
        elif split or layerwise:
            kind = "bf16x3 split on v_mfma_f32_32x32x16_bf16 (6 partial products)" + (", layer by layer" if layerwise else "")
            peak = PEAK_BF16_MFMA_TFLOPS / 6.0
        else:
            kind, peak = "fp32 on v_mfma_f32_32x32x2_f32", PEAK_FP32_MFMA_TFLOPS
        rows.append(dict(stage=stage, chain=name, dims=list(dims), flops_per_frame=fl, arithmetic=kind, peak_tflops=peak))

    for li, mod in enumerate(net.SA_modules):
        m = int(mod.npoint * scale)
        for si, (grouper, mlp) in enumerate(zip(mod.groupers, mod.mlps)):
            convs = [c for c in mlp.modules() if isinstance(c, torch.nn.Conv2d)]
            dims = [convs[0].in_channels] + [c.out_channels for c in convs]
            add("sa_mlp", "SA%d.%d" % (li, si), dims, m * grouper.nsample, True, dims[0] - 3, 0, grouper.nsample)
    n_unknown = [int(v * scale) for v in (12288, 2048, 1024, 512)]
    skip = [6, 96, 256, 512]
    for fi, mod in enumerate(net.FP_modules):
        convs = [c for c in mod.mlp.modules() if isinstance(c, torch.nn.Conv2d)]
        dims = [convs[0].in_channels] + [c.out_channels for c in convs]
        add("fp_mlp", "FP%d" % fi, dims, n_unknown[fi], False, dims[0] - skip[fi], skip[fi], 0)
    return rows


def run_postproc(inp, timer, poll_every):
    """(B) vote -> MeanShift x (K+1) -> Kabsch for the whole batch."""
    from pvn3d_amd.lib.utils import pvn3d_eval_utils as ev
    t = timer.start("vote_cluster_pose")
    res = ev.cal_batch_poses_lm(inp["pcld"], inp["mask"], inp["ctr_of"].unsqueeze(1) if inp["ctr_of"].dim() == 3
                                else inp["ctr_of"], inp["pred_kp_of"], True, 2, False, 1, poll_every=poll_every)
    timer.stop(t)
    return res


def _time_dense_frame(frame, threads, budget_s):
    """Dense torch-CPU restatement of the reference's post-processing on one frame with `threads` torch
    threads.  Returns (seconds per frame, seconds inside MeanShift fits, fits timed, extrapolated?)."""
    from oracle import posecal, torch_port
    torch.set_num_threads(threads)
    t_fit = []

    class _Stop(Exception):
        pass

    def fit(A, bw):
        t0 = time.perf_counter()
        c, l, it = torch_port.meanshift_fit_dense(torch.from_numpy(np.ascontiguousarray(A)), bw)
        t_fit.append(time.perf_counter() - t0)
        if len(t_fit) >= 2 and sum(t_fit) / len(t_fit) * 9 > budget_s:
            raise _Stop()
        return c.numpy(), l.numpy(), it
    t0 = time.perf_counter()
    try:
        posecal.cal_frame_poses_lm(frame["pcld"], frame["mask"], frame["ctr_of"], frame["pred_kp_of"], True, 2,
                                   False, frame["mesh_kps"], fit=fit, bft=torch_port.best_fit_transform_np)
        return time.perf_counter() - t0, sum(t_fit), len(t_fit), False
    except _Stop:
        per = sum(t_fit) / len(t_fit)
        return per * 9, per * 9, len(t_fit), True


def cpu_baseline(frame, budget_s=20.0, one_thread_budget_s=8.0):
    """Reference CPU path (dense torch MeanShift + numpy Kabsch, oracle/torch_port.py) on ONE frame of the
    same workload on the host cores, at the chosen thread count and at 1 thread; plus the C/OpenMP oracle.
    Bounded: if the first fits predict more than the budget for the frame, only those fits are run and
    the 9-fit frame time is extrapolated (and labelled)."""
    from oracle import posecal, native
    cores = os.cpu_count() or 1
    # the dense (n,n,3) temporaries are memory-bound: 32 threads measured 13x faster than 256 on the
    # 256-core GPU-box host (round 1), so the thread count is capped there
    threads = min(cores, 32)
    t_ref, t_ms, n_fits, extrapolated = _time_dense_frame(frame, threads, budget_s)
    t1, t1_ms, n1, ex1 = _time_dense_frame(frame, 1, one_thread_budget_s)
    torch.set_num_threads(threads)
    native.set_num_threads(min(cores, 64))
    t0 = time.perf_counter()
    posecal.cal_frame_poses_lm(frame["pcld"], frame["mask"], frame["ctr_of"], frame["pred_kp_of"], True, 2, False,
                               frame["mesh_kps"])
    t_c = time.perf_counter() - t0
    n_obj = int((frame["mask"] == 1).sum())
    sample = ("1 frame (N=%d, n_obj=%d, 9 fits) vote+cluster+pose; kind 'port' = oracle/torch_port.py, the dense "
              "torch-CPU restatement of MeanShiftTorch.fit (+ numpy Kabsch), pinned to the reference's own outputs "
              "(tests/golden/meanshift_ref.npz) and timed at 0.68-1.02x the reference's own fit on the same inputs "
              "(profiles/r04_meanshift_ref_vs_port.json, 8 threads, build container; /root/reference is absent on the "
              "bench host)" % (len(frame["pcld"]), n_obj))
    if extrapolated:
        sample += "; %d of 9 fits timed, frame time extrapolated x9/%d" % (n_fits, n_fits)
    return dict(value=1.0 / t_ref, unit="frames/s", cores=threads, kind="port", sample=sample,
                seconds_per_frame=t_ref, meanshift_share=t_ms / t_ref, host_cores_available=cores,
                threads_choice="min(cores, 32): the dense (n,n,3) temporaries are memory-bound; 256 threads "
                               "measured 13x slower than 32 on this host class",
                one_thread=dict(value=1.0 / t1, unit="frames/s", cores=1, seconds_per_frame=t1,
                                fits_timed=n1, extrapolated=ex1),
                c_openmp_port=dict(value=1.0 / t_c, unit="frames/s", cores=min(cores, 64),
                                   seconds_per_frame=t_c))


def gather_step_results(res, frames_local, frames_total, world, strong):
    """The path's only collective: the per-frame result rows (3x4 pose, K+1 keypoints, iteration counts =
    48 floats per frame) of every rank to every rank, once per batch (RCCL over xGMI on the GPU box, gloo in
    tests/test_sharding_gloo.py; no-op at world size 1).  strong: ranks own contiguous, possibly unequal
    blocks of `frames_total` frames (sharding.gather_frame_results -> one (frames_total, D) tensor in frame
    order); weak: every rank owns `frames_local` frames (plain all_gather -> list of per-rank tensors)."""
    if world == 1:
        return None
    from pvn3d_amd import sharding
    rows = torch.cat([res["poses"].reshape(frames_local, -1).to(torch.float32),
                      res["cls_kps"].reshape(frames_local, -1).to(torch.float32),
                      res["iters"].reshape(frames_local, -1).to(torch.float32)], 1)
    if strong:
        return sharding.gather_frame_results(rows, frames_total)
    bufs = [torch.empty_like(rows) for _ in range(world)]
    dist.all_gather(bufs, rows)
    return bufs


def _median_ms(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def device_copy_bandwidth(dev):
    """Achievable HBM bandwidth on this device: 1 GiB device-to-device copy (read + write) and fill."""
    a = torch.empty(256 * 1024 * 1024, device=dev)
    b = torch.empty_like(a)
    ms_c = _median_ms(lambda: b.copy_(a), 5)
    ms_f = _median_ms(lambda: b.fill_(1.0), 5)
    del a, b
    return dict(copy_gbs=2 * (1 << 30) / (ms_c * 1e-3) / 1e9, fill_gbs=(1 << 30) / (ms_f * 1e-3) / 1e9,
                unit="GB/s", bytes=1 << 30, note="copy counts read + write bytes")


def graph_pipeline_ms(net, dev, frames, steps=40, seed_base=7100):
    """A stream of `frames`-frame batches through lib/pipeline.py::GraphedPipeline (one HIP-graph replay per batch: feature
    path of batch i || xyz-only geometry of batch i+1 || vote -> cluster -> pose of batch i).  Four different batches take
    turns; the first round is checked against the eager calls (features and poses bit for bit).  -> dict."""
    from pvn3d_amd.lib.pipeline import GraphedPipeline
    off = StageTimer(False)
    batches = []
    for s in range(4):
        b = make_inputs(frames, 12288, 3072, dev, seed_base=seed_base + 100 * s)
        b["pc"] = torch.cat([b["pcld"], b["feats"].transpose(1, 2)], 2).contiguous()
        batches.append(b)
    post = lambda b: (b["pcld"], b["mask"], b["ctr_of"].unsqueeze(1) if b["ctr_of"].dim() == 3 else b["ctr_of"], b["pred_kp_of"])
    pipe = GraphedPipeline(net, batches[0]["pc"], post=post(batches[0]), obj_id=1)
    same = True
    for s, b in enumerate(batches):
        feats, res = pipe(b["pc"], pc_next=batches[(s + 1) % 4]["pc"], pc_next2=batches[(s + 2) % 4]["pc"], post=post(b))
        feats, poses = feats.clone(), res["poses"].clone()
        with torch.no_grad():
            want = net(b["pc"])
        same = same and bool(torch.equal(feats, want)) and bool(torch.equal(poses, run_postproc(b, off, 4)["poses"]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        b = batches[k % 4]
        pipe(b["pc"], pc_next=batches[(k + 1) % 4]["pc"], pc_next2=batches[(k + 2) % 4]["pc"], post=post(b))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return dict(ms_per_step=ms, frames_per_s=frames * 1e3 / ms, identical_to_eager_calls=same, fallbacks=pipe.fallbacks,
                note="lib/pipeline.py::GraphedPipeline (depth 3): one graph replay per batch -- the first-level FPS run of the "
                     "batch after next, the next batch's ball queries / three_nn and this batch's MLP kernels and vote stage "
                     "beside each other; bounded MeanShift iterations, one host read per call")


def per_rank_share_entry(net, dev, poll_every, frames=8, steps=20, warm=5):
    """What one rank of BASELINE config 4 runs: `frames` frames per step through the headline's pipelined step (MLP feature
    path on a side stream with the NEXT step's xyz-only geometry enqueued ahead of it, vote -> cluster -> pose on the
    current stream), plus the serial stage times of the same step.  At 8 frames the step is latency-bound: FPS level 0 is
    one wave per cloud (1.53 ms whatever the frame count) and the MeanShift iterations are a handful of waves."""
    off = StageTimer(False)
    inp = make_inputs(frames, 12288, 3072, dev, seed_base=7050)
    inp["pc"] = torch.cat([inp["pcld"], inp["feats"].transpose(1, 2)], 2).contiguous()
    side = torch.cuda.Stream(device=dev)
    geo_next = [None]

    def step():
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            with torch.no_grad():
                geo = geo_next[0] if geo_next[0] is not None else net.geometry_ahead(inp["pc"])
                geo_next[0] = net.geometry_ahead(inp["pc"])
            keep = run_net(net, inp, off, geometry=geo)
        res = run_postproc(inp, off, poll_every)
        torch.cuda.current_stream(dev).wait_stream(side)
        return keep, res

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        keep, res = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    timer = StageTimer(True)
    for _ in range(steps):
        run_net(net, inp, timer)
        run_postproc(inp, timer, poll_every)
    torch.cuda.synchronize()
    stage = {k: v / steps for k, v in timer.totals_ms().items()}
    p = res["poses"].cpu().numpy()
    err = float(max(max(np.abs(p[i][:, :3] - f["R"]).max(), np.abs(p[i][:, 3] - f["t"]).max()) for i, f in enumerate(inp["frames"])))
    return dict(name="config4_per_rank_share", workload="BASELINE config 4 as ONE of its 8 ranks runs it: %d frames per step "
                "(64 frames sharded 8-way), N=12288, n_obj=3072, K=8; the headline's three-stream pipelined step" % frames,
                frames_per_step=frames, ms_per_step=ms, frames_per_s=frames * 1e3 / ms, stage_ms_per_step_serial=stage,
                graph_pipeline=graph_pipeline_ms(net, dev, frames),
                implied_8_gpu_frames_per_s_if_ranks_do_not_interfere=8 * frames * 1e3 / ms,
                note="the 8-GPU figure is arithmetic on a 1-GPU measurement (the driver measures the real curve); the only "
                     "cross-rank step is one all-gather of 48 floats per frame",
                pose_err_vs_ground_truth=err)


def extra_configs(net, dev, poll_every, with_cpu):
    """The BASELINE configurations next to the headline one, each on a few steps (rank 0, N = 1)."""
    from pvn3d_amd import synth
    from pvn3d_amd.lib.utils import pvn3d_eval_utils as ev
    out = []
    off = StageTimer(False)

    def to_dev(frames):
        st = lambda k: torch.from_numpy(np.stack([f[k] for f in frames], 0)).to(dev)
        return dict(pcld=st("pcld").contiguous(), mask=st("mask").to(torch.int32).contiguous(),
                    ctr_of=st("ctr_of").contiguous(), pred_kp_of=st("pred_kp_of").contiguous(), frames=frames)

    def post(inp):
        return run_postproc(inp, off, poll_every)

    def pose_err(res, frames):
        p = res["poses"].cpu().numpy()
        return float(max(max(np.abs(p[i][:, :3] - f["R"]).max(), np.abs(p[i][:, 3] - f["t"]).max())
                         for i, f in enumerate(frames)))

    # (i) B = 1 latency, config 2 (the reference evaluates at test_mini_batch_size = 1, common.py:41)
    inp = make_inputs(1, 12288, 3072, dev, seed_base=7000)
    inp["pc"] = torch.cat([inp["pcld"], inp["feats"].transpose(1, 2)], 2).contiguous()
    ms_a = _median_ms(lambda: run_net(net, inp, off), 20) if net is not None else None
    ms_b = _median_ms(lambda: post(inp), 20)
    both = (lambda: (run_net(net, inp, off), post(inp))) if net is not None else (lambda: post(inp))
    ms_ab = _median_ms(both, 20)
    ms_g = ms_gab = ms_fps = None
    if net is not None:
        # the same forward as one HIP-graph replay (shapes are static at N = 12288), and FPS level 0 alone
        from pvn3d_amd.lib.pointnet2_utils import _ext as _e
        with torch.no_grad():
            want = net(inp["pc"]).clone()
        g = net.graphed(inp["pc"])
        assert torch.equal(g(inp["pc"]), want), "graph replay differs from the eager forward"
        ms_g = _median_ms(lambda: g(inp["pc"]), 20)
        ms_gab = _median_ms(lambda: (g(inp["pc"]), post(inp)), 20)
        ms_fps = _median_ms(lambda: _e.furthest_point_sampling(inp["pcld"], 2048), 20)
    # vote -> cluster -> pose of the frame as one HIP-graph replay too (bounded MeanShift iterations, no host poll)
    from pvn3d_amd.lib.utils import pvn3d_eval_utils as _ev
    ctr1 = inp["ctr_of"].unsqueeze(1) if inp["ctr_of"].dim() == 3 else inp["ctr_of"]
    gp = _ev.GraphedFramePoses("lm", inp["pcld"], inp["mask"], ctr1, inp["pred_kp_of"], n_cls=2, obj_id=1,
                               use_ctr_clus_flter=False)
    ms_bg = _median_ms(lambda: gp(inp["pcld"], inp["mask"], ctr1, inp["pred_kp_of"]), 20)
    ms_gg = None
    if net is not None:
        ms_gg = _median_ms(lambda: (g(inp["pc"]), gp(inp["pcld"], inp["mask"], ctr1, inp["pred_kp_of"])), 20)
        del g
    res = post(inp)
    out.append(dict(name="b1_latency", workload="config 2 (LineMOD eval path), ONE frame per call: N=12288, n_obj=3072, K=8",
                    ms_per_frame=dict(pointnet2_msg=ms_a, vote_cluster_pose=ms_b, both_serial=ms_ab,
                                      pointnet2_msg_graph=ms_g, both_serial_graph=ms_gab, fps_level0=ms_fps,
                                      vote_cluster_pose_graph=ms_bg, both_serial_both_graphs=ms_gg),
                    frames_per_s=1e3 / ms_ab, meanshift_iters_max=int(res["iters"].max().item()),
                    stream_of_single_frames=graph_pipeline_ms(net, dev, 1, seed_base=7200) if net is not None else None,
                    pose_err_vs_ground_truth=pose_err(res, inp["frames"])))

    # (i') BASELINE config 4's per-rank share: 64 frames sharded 8-way = 8 frames per GPU per step
    # (TorchEval.eval_pose_parallel, pvn3d_eval_utils.py:345-387, is the contract the step replaces): the same three-stream
    # step as the headline (feature path || geometry of the next batch || vote-cluster-pose), 8 frames
    if net is not None:
        out.append(per_rank_share_entry(net, dev, poll_every, frames=8))
        # (i'') frames per step against throughput on the graph pipeline (1 and 8 frames are in the two entries above, 64 is
        # the headline's eager step): where the FPS chain stops and the two compute islands start to bound the step
        sweep = {}
        for fr_n in (16, 32):
            r = graph_pipeline_ms(net, dev, fr_n, steps=20, seed_base=7300 + fr_n)
            sweep[str(fr_n)] = dict(ms_per_step=r["ms_per_step"], frames_per_s=r["frames_per_s"],
                                    identical_to_eager_calls=r["identical_to_eager_calls"])
        out.append(dict(name="graph_pipeline_batch_sweep", workload="config 2 frames through lib/pipeline.py::GraphedPipeline, "
                        "16 and 32 frames per step (N=12288, n_obj=3072, K=8)", frames_per_step=sweep))

    # (ii) the n_obj = 12 288 stress case (every point of the cloud votes), 8 frames per call
    fr = [synth.synth_frame(frame=7100 + i, n_pts=12288, n_obj=12288) for i in range(8)]
    inp = to_dev(fr)
    ms = _median_ms(lambda: post(inp), 5)
    res = post(inp)
    it = res["iters"].cpu().numpy().astype(np.float64)
    pairs = float((it * 12288.0 * 12288.0).sum())
    out.append(dict(name="stress_nobj_12288", workload="vote -> MeanShift x9 -> Kabsch with n_obj = N = 12288, 8 frames per call",
                    ms_per_frame=ms / 8, frames_per_s=8e3 / ms, meanshift_iters=dict(min=int(it.min()), max=int(it.max()), mean=float(it.mean())),
                    pair_evals_per_s=pairs / (ms * 1e-3), pose_err_vs_ground_truth=pose_err(res, fr)))

    # (iii) config 3: YCB multi-instance frame, 21 classes, 5 objects, centre-cluster filter on
    fy = [synth.synth_frame_ycb(frame=7200 + i) for i in range(16)]
    sty = lambda k: torch.from_numpy(np.stack([f[k] for f in fy], 0)).to(dev)
    yp, ym, yc, yk = sty("pcld").contiguous(), sty("mask").to(torch.int32).contiguous(), sty("ctr_of").contiguous(), sty("pred_kp_of").contiguous()
    run_y = lambda: ev.cal_batch_poses(yp, ym, yc, yk, True, 22, True, poll_every=poll_every)
    ms = _median_ms(run_y, 5)
    ry = run_y()
    poses = ry["poses"].cpu().numpy()
    err = 0.0
    for i, f in enumerate(fy):
        for cid, (R, t) in f["poses"].items():
            err = max(err, float(np.abs(poses[i, cid - 1][:, :3] - R).max()), float(np.abs(poses[i, cid - 1][:, 3] - t).max()))
    ms1 = _median_ms(lambda: ev.cal_batch_poses(yp[:1], ym[:1], yc[:1], yk[:1], True, 22, True, poll_every=poll_every), 10)
    # the same single-frame call as one HIP-graph replay (bounded MeanShift iterations, no host poll; frames that do not
    # finish within the bound are repeated through the polled call -- none here)
    gy = ev.GraphedFramePoses("ycb", yp[:1], ym[:1], yc[:1], yk[:1], n_cls=22)
    ms1g = _median_ms(lambda: gy(yp[:1], ym[:1], yc[:1], yk[:1]), 10)
    out.append(dict(name="ycb_multi_instance", workload="config 3: N=12288, 21 classes, 5 objects of 1228 points, use_ctr_clus_flter=True; "
                                                         "16 frames per call",
                    ms_per_frame=ms / 16, frames_per_s=16e3 / ms, ms_single_frame_call=ms1,
                    ms_single_frame_call_graph=ms1g, graph_fallbacks=gy.fallbacks,
                    meanshift_iters_max=int(ry["iters"].max().item()), pose_err_vs_ground_truth=err))

    # (iv) config 1: one 2048-point cloud, 1 object, all points on it; CPU path in full
    f1 = [synth.synth_frame(frame=7300 + i, n_pts=2048, n_obj=2048) for i in range(64)]
    inp1, inp64 = to_dev(f1[:1]), to_dev(f1)
    ms1 = _median_ms(lambda: post(inp1), 20)
    ms64 = _median_ms(lambda: post(inp64), 5)
    c1 = dict(name="config1_n2048", workload="config 1: single synthetic 2048-pt cloud, 1 object, 8 keypoints (+centre): "
                                            "vote -> MeanShift x9 -> Kabsch",
              ms_per_frame_b1=ms1, ms_per_frame_b64=ms64 / 64, frames_per_s_b64=64e3 / ms64,
              pose_err_vs_ground_truth=pose_err(post(inp64), f1))
    if with_cpu:
        cores = os.cpu_count() or 1
        t_ref, t_ms, _n, _e = _time_dense_frame(f1[0], min(cores, 32), 1e9)
        t_one, _a, _b, _c = _time_dense_frame(f1[0], 1, 1e9)
        c1["cpu_reference_path"] = dict(seconds_per_frame=t_ref, cores=min(cores, 32), seconds_per_frame_1_thread=t_one,
                                        meanshift_share=t_ms / t_ref, kind="port",
                                        sample="1 frame in full (9 fits), dense torch-CPU MeanShift + numpy Kabsch")
        c1["gpu_over_cpu_b1"] = t_ref * 1e3 / ms1
    out.append(c1)

    # (v) heavy-tailed votes: 10 % outliers with sigma = 30 cm (far points creep for many iterations)
    fh = [synth.synth_frame(frame=7400 + i, n_pts=12288, n_obj=3072, sig_out=0.30) for i in range(16)]
    inph = to_dev(fh)
    ms = _median_ms(lambda: post(inph), 5)
    res = post(inph)
    it = res["iters"].cpu().numpy().astype(np.float64)
    cnt = res["counts"].cpu().numpy().astype(np.float64)
    # the same call with the reference's stop rule run to the end (no winner stop): the iteration counts the reference
    # would run on these votes, the time they cost, and that the poses are the same bits
    from pvn3d_amd.lib.utils import _vote_engine
    saved = _vote_engine.DEFAULT_KERNEL
    _vote_engine.DEFAULT_KERNEL = "nowin"
    try:
        ms_full = _median_ms(lambda: post(inph), 3)
        res_full = post(inph)
    finally:
        _vote_engine.DEFAULT_KERNEL = saved
    itf = res_full["iters"].cpu().numpy().astype(np.float64)
    out.append(dict(name="heavy_tail_votes", workload="config 2 frames with 10 % vote outliers of sigma = 30 cm, 16 frames per call",
                    ms_per_frame=ms / 16, frames_per_s=16e3 / ms,
                    meanshift_iters=dict(min=int(it.min()), max=int(it.max()), mean=float(it.mean())),
                    meanshift_iters_reference_stop_rule=dict(min=int(itf.min()), max=int(itf.max()), mean=float(itf.mean())),
                    ms_per_frame_reference_stop_rule=ms_full / 16,
                    poses_identical_to_reference_stop_rule=bool(torch.equal(res["poses"], res_full["poses"])),
                    valu_tflops_16flop_per_pair=16.0 * float((it * cnt * cnt).sum()) / (ms * 1e-3) / 1e12,
                    pose_err_vs_ground_truth=pose_err(res, fh)))
    # (vi) config 5: one training step of the voting branch, mini_batch_size = 24 (common.py:37), fp32 and bf16
    from pvn3d_amd import train_step as ts
    B = 24
    batch = ts.synthetic_batch(B, 12288, dev, seed_base=7500, n_obj=3072)
    from pvn3d_amd.lib.pointnet2_utils import _train_mlp
    entry = dict(name="train_step", workload="config 5 on ONE GPU: Pointnet2MSG + offset heads, forward + vote loss + backward + Adam "
                                              "step, %d frames of N=12288 per step; native gather/scatter ops + vote loss; "
                                              "bf16: SA/FP SharedMLP forward+backward on csrc/mlp_train.hip (bf16 MFMA GEMM + fused "
                                              "BatchNorm/ReLU/pool kernels); bf16_library_mlp / fp32: torch Conv2d/BatchNorm2d "
                                              "(MIOpen/hipBLASLt)" % B)
    # bf16_autocast_prefetch: the same step with the NEXT batch's xyz-only geometry (FPS, ball query, three_nn) enqueued
    # under this step's backward, as a training loop with a data loader can (here the next batch is the same tensor)
    for tag, dt, fused, pre in (("fp32", None, False, False), ("bf16_library_mlp", torch.bfloat16, False, False),
                                ("bf16_autocast", torch.bfloat16, True, False),
                                ("bf16_autocast_prefetch", torch.bfloat16, True, True)):
        torch.manual_seed(1)
        model = ts.PointVoteNet().to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        losses = []
        _train_mlp.TRAIN_FUSED = fused
        try:
            ms = _median_ms(lambda: losses.append(ts.train_step(model, opt, batch, autocast_dtype=dt,
                                                                prefetch=batch["pc"] if pre else None)), 5, warm=2)
        finally:
            _train_mlp.TRAIN_FUSED = "auto"
        entry[tag] = dict(ms_per_step=ms, frames_per_s=B * 1e3 / ms, loss_first=float(losses[0]), loss_last=float(losses[-1]),
                          peak_mem_gb=torch.cuda.max_memory_allocated(dev) / 2 ** 30)
        if tag == "bf16_autocast":
            alg = ts.algorithmic_work_per_step(model.backbone, B)
        del model, opt
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
    # roofline of the step's SA / FP SharedMLP chains (what csrc/mlp_train.hip owns; FPS, the heads, the loss and Adam
    # are in the step time but not in the algorithmic work): both bounds, the binding one first
    t = entry["bf16_autocast"]["ms_per_step"] * 1e-3
    t_hbm, t_mfma = alg["bytes"] / (PEAK_HBM_GBS * 1e9), alg["flops"] / (PEAK_BF16_MFMA_TFLOPS * 1e12)
    rl = dict(bound="hbm" if t_hbm >= t_mfma else "mfma", achieved=alg["bytes"] / t / 1e9, peak=PEAK_HBM_GBS, unit="GB/s",
              frac=alg["bytes"] / t / 1e9 / PEAK_HBM_GBS, traffic=None,
              algorithmic_bytes_per_step=alg["bytes"], algorithmic_flops_per_step=alg["flops"],
              mfma_view=dict(achieved=alg["flops"] / t / 1e12, peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s",
                             frac=alg["flops"] / t / 1e12 / PEAK_BF16_MFMA_TFLOPS),
              time_at_hbm_peak_ms=t_hbm * 1e3, time_at_mfma_peak_ms=t_mfma * 1e3,
              note="denominator = the whole bf16_autocast step (incl. FPS, heads, loss, Adam); bytes = bf16 matrices "
                   "the per-layer passes must stream (train_step.algorithmic_work_per_step); the narrow-K GEMMs make "
                   "HBM, not MFMA, the binding roofline")
    try:
        with open(os.path.join(ROOT, "profiles", "latest_train_pmc_traffic.json")) as f:
            pmc = json.load(f)
        if pmc.get("frames_per_step") == B:
            rl["traffic"] = pmc["sa_fp_mlp_chain_bytes_per_step"]
            rl["traffic_all_kernels"] = pmc["all_kernels_bytes_per_step"]
            rl["traffic_source"] = "profiles/%s_train_pmc_traffic.json" % pmc.get("tag")
            rl["traffic_measured_in_run"] = False
    except (OSError, ValueError, KeyError):
        pass
    entry["roofline"] = rl
    out.append(entry)
    return out


def mlp_arithmetic_field():
    """config.arithmetic: what the SA / FP contractions compute in."""
    from pvn3d_amd.lib.pointnet2_utils import _fused_mlp
    a = _fused_mlp.MLP_ARITH
    return {"name": a,
            "meaning": {"fp16x2": "fp32 operands as two fp16 pieces each, three exact partial products per multiply on "
                                  "v_mfma_f32_32x32x16_f16, fp32 accumulation; every chain rescaled diagonally by powers of "
                                  "two on the host (per-row weight scales, per-channel hidden bounds), activations scaled from "
                                  "device-side bounds; a chain the host-side probe finds unsafe runs bf16x3",
                        "bf16x3": "three bf16 pieces per operand, six partial products on v_mfma_f32_32x32x16_bf16",
                        "fp32": "v_mfma_f32_32x32x2_f32"}.get(a, a),
            "operand_bits": {"fp16x2": 22, "bf16x3": 24, "fp32": 24}.get(a)}


def distributed_train_entry(dev, rank, world, steps=5, warm=2, frames=24, bucket_bytes=4 << 20, standin=False):
    """BASELINE config 5 at N > 1 (weak: `frames` frames per GPU): the bf16 training step of the voting branch with its
    gradient buckets all-reduced from inside backward (sharding.OverlappedGradientReducer over RCCL / xGMI) -- what
    replaces the reference's nn.DataParallel (train_linemod_pvn3d.py:480,502-503).  Every rank calls this; the timing
    is bracketed by barrier + synchronize on both sides and the MAX over ranks is reported.  The same K steps are then
    repeated WITHOUT the exchange (group of one) so that the exposed all-reduce time -- what the collectives add to a
    step after their overlap with backward -- is a measured difference, not an estimate.
    standin=True (CPU test of this code path only, PVN3D_BENCH_TRAIN_STANDIN=1 under the gloo override): a small
    torch-only model on CPU tensors goes through the SAME exchange / timing / reporting code; labelled in the entry."""
    from pvn3d_amd import sharding
    group = dist.group.WORLD
    if standin:
        torch.manual_seed(3)
        model = torch.nn.Sequential(torch.nn.Linear(64, 512), torch.nn.ReLU(), torch.nn.Linear(512, 512), torch.nn.ReLU(),
                                    torch.nn.Linear(512, 27))
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        torch.manual_seed(100 + rank)
        x, y = torch.randn(frames * 64, 64), torch.randn(frames * 64, 27)
        bucket_bytes = 256 << 10

        def one_step(g):
            opt.zero_grad(set_to_none=True)
            loss = ((model(x) - y) ** 2).mean()
            red = sharding.overlapped_reducer(model, bucket_bytes=bucket_bytes, group=g) if g is not None else None
            if red is not None:
                red.arm()
            loss.backward()
            if red is not None:
                red.finalize()
            opt.step()
            return loss.detach()
        sync = lambda: None
        net = model
        what = "STAND-IN (3-layer torch MLP on CPU tensors): exercises the launch / exchange / report path only"
    else:
        from pvn3d_amd import train_step as ts
        torch.manual_seed(1)
        model = ts.PointVoteNet().to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        batch = ts.synthetic_batch(frames, 12288, dev, seed_base=7500 + 1000 * rank, n_obj=3072)
        # (g None = the baseline leg: exchange switched off explicitly -- group=None alone would mean "the default group")
        one_step = lambda g: ts.train_step(model, opt, batch, autocast_dtype=torch.bfloat16, group=group,
                                           bucket_bytes=bucket_bytes, exchange=g is not None)
        sync = lambda: torch.cuda.synchronize(dev)
        net = model
        what = ("Pointnet2MSG + offset heads, forward + vote loss + backward + Adam step, bf16 autocast, SA/FP SharedMLP "
                "on csrc/mlp_train.hip; %d frames of N=12288 per GPU per step" % frames)
    sharding.broadcast_parameters(net, group=group)          # every rank starts from rank 0's weights (DataParallel's replication)

    def timed(g):
        for _ in range(warm):
            one_step(g)
        sync()
        dist.barrier()
        sync()
        t0 = time.perf_counter()
        losses = [one_step(g) for _ in range(steps)]
        sync()
        dist.barrier()
        sync()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if not standin else "cpu")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()) / steps * 1e3, [float(l) for l in losses]

    ms_x, losses = timed(group)
    red = getattr(net, "_pvn3d_grad_reducer", None)
    n_buckets = len(red.buckets) if red is not None else 0
    bytes_buckets = [int(sum(p.numel() * p.element_size() for p in b)) for b in red.buckets] if red is not None else []
    during = red.launched_during_backward if red is not None else 0
    # weights identical on every rank after the exchanged steps (the whole point of the all-reduce)
    chk = torch.cat([p.detach().reshape(-1)[:64].float().cpu() for p in net.parameters()])
    lo, hi = chk.clone(), chk.clone()
    if not standin:
        lo, hi = lo.to(dev), hi.to(dev)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    in_sync = bool(torch.equal(lo, hi))
    issued_before = red.collectives_issued if red is not None else 0
    ms_0, _ = timed(None)                                    # the same steps, no exchange
    baseline_collectives = (red.collectives_issued if red is not None else 0) - issued_before
    if baseline_collectives != 0:
        raise RuntimeError("the no-exchange leg issued %d gradient all-reduces" % baseline_collectives)
    return dict(name="train_step", n_gpus=world, scaling="weak", frames_per_gpu_per_step=frames,
                workload="config 5 at %d ranks: %s; gradients bucketed (%.1f MiB) and all-reduced from inside backward "
                         "(RCCL over xGMI), one process per GPU" % (world, what, bucket_bytes / 2 ** 20),
                ms_per_step=ms_x, frames_per_s=world * frames * 1e3 / ms_x,
                ms_per_step_without_exchange=ms_0, exposed_allreduce_ms=max(0.0, ms_x - ms_0),
                gradient_allreduces_in_the_no_exchange_leg=int(baseline_collectives),
                gradient_buckets=n_buckets, gradient_bytes_per_step=int(sum(bytes_buckets)), bucket_bytes=bytes_buckets,
                buckets_issued_inside_backward_total=int(during), steps=steps, warmup=warm,
                weights_identical_across_ranks_after_steps=in_sync, loss_first=losses[0], loss_last=losses[-1],
                backend=dist.get_backend())


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: re-execute this command line as N ranks (one per GPU) under
    torch.distributed.run on 127.0.0.1 with a free port, stream rank 0's JSON line through, return the exit code.
    (The driver's own form, `python -m torch.distributed.run ... bench.py --gpus N`, sets WORLD_SIZE and never comes
    here.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--n-pts", type=int, default=12288)
    ap.add_argument("--n-obj", type=int, default=3072)
    ap.add_argument("--poll-every", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-events", action="store_true")
    ap.add_argument("--serial", action="store_true", help="one stream, islands back to back")
    ap.add_argument("--ms-kernel", default="sgpr",
                    help="MeanShift iteration kernel for the timed steps: 'sgpr' (default here: the LDS-free kernel built to "
                         "run beside the MLP kernels, bit-identical results), 'sgpr+cap1024', 'packed+split', ...; "
                         "'library' = the library's own default (the LDS kernel)")
    ap.add_argument("--geometry-ahead", action="store_true", help="(default since round 5; kept for old command lines)")
    ap.add_argument("--no-geometry-ahead", action="store_true",
                    help="every step stands alone: its xyz-only work (FPS, ball query, three_nn) heads its own MLP stream. "
                         "Default: a pipelined evaluator -- step i enqueues the geometry of step i+1's batch beside its own MLP "
                         "kernels (round 5, with the MLP chains no longer matrix-bound: 11.26 -> 10.56 ms per step; round 4: 1 %)")
    ap.add_argument("--ops-only", action="store_true",
                    help="island (A) = bare SA/FP op chain with synthetic features (no MLP GEMMs)")
    ap.add_argument("--strong", action="store_true",
                    help="--frames is the TOTAL per step, split over the ranks (BASELINE config 4: 64 frames sharded)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the `configs` array (N=1 only)")
    ap.add_argument("--train-only", action="store_true",
                    help="N > 1: only BASELINE config 5 (distributed_train_entry), one JSON line with that entry")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launch + init_process_group + one all-reduce, then exit (checks the N > 1 launch path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run, same flags
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d (launch with --nproc-per-node == --gpus, or without a "
                         "launcher: bench.py starts its own ranks)" % (world, args.gpus))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL needs it on this host driver)
        # "nccl" IS RCCL on ROCm; the override exists for the CPU test of the launch path (gloo, no GPU)
        dist.init_process_group(os.environ.get("PVN3D_BENCH_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    if args.rendezvous_only:
        t = torch.tensor([float(rank + 1)])
        if world > 1:
            if dist.get_backend() == "nccl":
                torch.cuda.set_device(local_rank)
                t = t.cuda()
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"rendezvous": world, "rank_sum": float(t.item()),
                              "backend": dist.get_backend() if world > 1 else None}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    standin = os.environ.get("PVN3D_BENCH_TRAIN_STANDIN") == "1"      # CPU test of the config-5 launch / exchange path
    if args.train_only:
        if world < 2:
            raise SystemExit("bench.py --train-only measures config 5 at N > 1 (the N = 1 step is in `configs` of the default run)")
        if not standin:
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the measured path)")
            torch.cuda.set_device(local_rank)
        entry = distributed_train_entry(torch.device("cuda", local_rank) if not standin else torch.device("cpu"), rank, world,
                                        steps=args.steps, warm=args.warmup, standin=standin)
        if rank == 0:
            print(json.dumps({"metric": "config 5 training step, %d ranks" % world, "n_gpus": world, "configs": [entry]}))
        dist.barrier()
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    scale = args.n_pts / 12288.0
    from pvn3d_amd import sharding
    if args.strong:
        lo, hi = sharding.shard_range(args.frames, rank, world)
        frames_local, frames_total = hi - lo, args.frames
        assert frames_local > 0, "--strong needs at least one frame per rank"
    else:
        frames_local, frames_total = args.frames, args.frames * world
    inp = make_inputs(frames_local, args.n_pts, args.n_obj, dev, seed_base=1000 * rank)
    inp["pc"] = torch.cat([inp["pcld"], inp["feats"].transpose(1, 2)], 2).contiguous()   # (F, N, 3+6)
    net = None if args.ops_only else make_net(dev)
    timer_off = StageTimer(False)

    def island_a(timer):
        return run_ops(inp, timer, scale) if net is None else run_net(net, inp, timer)

    # The two halves of the path are independent islands (the CNN + heads
    # sit between them), so a pipelined evaluator runs them concurrently on different frames.
    # Island (A) goes to a side HIP stream, island (B) stays on the current one; with
    # --serial both run back to back on one stream (per-stage event timings are taken in that
    # mode so they do not overlap).
    side = torch.cuda.Stream(device=dev)          # (a high-priority side stream changes nothing: 10.42 vs 10.38 ms)
    if args.ms_kernel and args.ms_kernel != "library":
        from pvn3d_amd.lib.utils import _vote_engine
        _vote_engine.DEFAULT_KERNEL = args.ms_kernel

    def gather_results(res):
        return gather_step_results(res, frames_local, frames_total, world, args.strong)

    # Software pipelining across steps (default since round 5; --no-geometry-ahead turns it off): a pipelined evaluator has
    # the next batch's cloud while it works on this one, so the xyz-only work of step i+1's batch (FPS, ball query,
    # three_nn: no features involved) is enqueued on the network's geometry stream BEFORE step i's MLP kernels, which then
    # start at once on the handle that step i-1 left.  Every step still enqueues exactly one geometry pass, one MLP pass
    # and one vote pass inside the timed region -- nothing is cached or skipped (the geometry is recomputed every step).
    # Round 4 measured 1 % for this (the MLP kernels were matrix-bound and the MeanShift iterations VALU-bound: moving
    # them beside each other only slowed both); with the chains at a third of their matrix-pipe time (round 5) the two
    # islands do overlap: 11.26 -> 10.8 ms, and 10.56 ms with the LDS-free MeanShift kernel.  The self-contained step is
    # measured right after the timed region and reported as `value_self_contained_steps`.
    geo_ahead = not args.no_geometry_ahead and net is not None and not args.serial
    geo_next = [None]

    def step(timer, ahead=None):
        ahead = geo_ahead if ahead is None else ahead
        if args.serial:
            keep = island_a(timer)
            res = run_postproc(inp, timer, args.poll_every)
            return keep, res, gather_results(res)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            if ahead:
                with torch.no_grad():
                    geo = geo_next[0] if geo_next[0] is not None else net.geometry_ahead(inp["pc"])
                    geo_next[0] = net.geometry_ahead(inp["pc"])
                keep = run_net(net, inp, timer_off, geometry=geo)
            else:
                keep = island_a(timer_off)
        res = run_postproc(inp, timer_off, args.poll_every)
        torch.cuda.current_stream(dev).wait_stream(side)
        return keep, res, gather_results(res)

    for _ in range(args.warmup):
        step(timer_off)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer = StageTimer(args.serial and not args.no_stage_events)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        keep, res, gathered = step(timer)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed_alone = None
    if geo_ahead:
        # the same K steps with every step standing alone (its own geometry at the head of its MLP stream)
        geo_next[0] = None
        step(timer_off, ahead=False)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(timer_off, ahead=False)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed_alone = time.perf_counter() - t1
    op_timer = StageTimer(False)
    if not args.no_stage_events:
        # per-stage kernel time: the same K steps once more, serialised on one stream so that the
        # event pairs bracket exactly one stage each (not part of `value`)
        if not args.serial:
            island_a(timer_off)                # warm this stream's allocator pool (untimed)
            torch.cuda.synchronize()
            timer = StageTimer(True)
            for _ in range(args.steps):
                island_a(timer)
                run_postproc(inp, timer, args.poll_every)
            torch.cuda.synchronize()
        if net is not None:
            # HBM rooflines of the data-movement ops: the unfused op chain at the same shapes
            run_ops(inp, timer_off, scale)
            torch.cuda.synchronize()
            op_timer = StageTimer(True)
            for _ in range(args.steps):
                run_ops(inp, op_timer, scale)
            torch.cuda.synchronize()
        else:
            op_timer = timer
    if world > 1:
        t = torch.tensor([elapsed, elapsed_alone if elapsed_alone is not None else 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        if elapsed_alone is not None:
            elapsed_alone = float(t[1].item())

    # sanity: the timed result is the real thing (pose close to the synthetic ground truth)
    pose0 = res["poses"][0].cpu().numpy()
    f0 = inp["frames"][0]
    pose_err = float(max(np.abs(pose0[:, :3] - f0["R"]).max(), np.abs(pose0[:, 3] - f0["t"]).max()))
    iters = res["iters"].cpu().numpy()

    # BASELINE config 5 at N > 1: every rank takes part (gradient all-reduce); reported by rank 0 in `configs`
    train_dist = None
    if world > 1 and not args.no_extra_configs and args.n_pts == 12288 and net is not None:
        del keep, gathered
        torch.cuda.empty_cache()
        train_dist = distributed_train_entry(dev, rank, world)
    if rank == 0:
        total_frames = frames_total * args.steps
        stage_ms = timer.totals_ms()
        per_step = {k: v / args.steps for k, v in stage_ms.items()}
        op_step = {k: v / args.steps for k, v in op_timer.totals_ms().items()}
        alg = algorithmic_bytes_per_frame(args.n_pts)
        rooflines = {}
        F = frames_local
        for name in ["ball_query", "group", "three_interpolate", "three_nn", "gather", "fps"]:
            if name in op_step and op_step[name] > 0:
                gbs = alg[name] * F / (op_step[name] * 1e-3) / 1e9
                rooflines[name] = dict(bound="hbm", achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s",
                                       frac=gbs / PEAK_HBM_GBS, traffic=None, ms_per_step=op_step[name],
                                       algorithmic_bytes_per_frame=alg[name])
        # The pair-scanning ops move kilobytes: what bounds them is VALU issue / latency.  The grid (ball query, three_nn)
        # and cell-culled (FPS) kernels EXECUTE a few per cent of the reference's brute-force pair evaluations (the rest is
        # excluded exactly), so a ratio of the reference's pair count to an issue-rate peak is not a roofline fraction
        # (it exceeded 1 in round 4): `achieved` is stated in reference pairs per second as a speed, `frac` is null, and
        # the HBM view above (algorithmic bytes / time) is the only fraction reported for them.
        pairs = pair_evals_per_frame(args.n_pts)
        for name in ("fps", "three_nn", "ball_query"):
            if name in op_step and op_step[name] > 0:
                gp = pairs[name] * F / (op_step[name] * 1e-3) / 1e9
                peak = VALU_LANE_INSTR_PER_S / VALU_INSTR_PER_PAIR[name] / 1e9
                entry = dict(bound="valu", bound_note="VALU issue / latency; exact culling: most reference pairs are never "
                             "evaluated, so no fraction of a brute-force bound is claimed", achieved=gp,
                             peak=None, unit="G reference-pairs/s", frac=None, traffic=None, ms_per_step=op_step[name],
                             pair_evals_per_frame_reference=pairs[name], valu_instr_per_pair=VALU_INSTR_PER_PAIR[name],
                             brute_force_issue_bound_gpairs=peak, speed_over_brute_force_issue_bound=gp / peak,
                             hbm_view=rooflines.get(name))
                if name == "fps":
                    entry["simds_in_use"] = F       # one wave per cloud (csrc/fps_cells.hip): F of the chip's 1024 SIMDs
                    entry["note"] = ("serial in the samples: one wave per cloud, 0.75 us per sampling round; the cell test "
                                     "touches ~2 of 64 cells per round (DESIGN 4.1)")
                rooflines[name] = entry
        if net is not None:
            sa_fl, fp_fl = mlp_flops_per_frame(net, scale)
            # Per chain: which matrix pipe it runs on.  fp32 chains: v_mfma_f32_32x32x2_f32, peak 157.3 TFLOP/s.  Split
            # chains (csrc/sa_mlp_split.hip): every fp32 multiply is six bf16 x bf16 partial products on
            # v_mfma_f32_32x32x16_bf16 (fp32 accuracy), so of the ALGORITHMIC fp32 flops the bf16 pipe
            # can deliver at most 2500 / 6 = 416.7 TFLOP/s.  A stage mixes both: its `peak` is the rate at which the
            # stage's chains would finish with every pipe at its dense peak (sum of flops / sum of ideal times), its
            # `frac` = ideal time / measured time -- never a ratio against a peak the launch does not run on.
            chains = mlp_chain_table(net, scale, F)
            for name, fl in (("sa_mlp", sa_fl), ("fp_mlp", fp_fl)):
                if per_step.get(name, 0) > 0:
                    mine = [c for c in chains if c["stage"] == name]
                    t_ideal = sum(c["flops_per_frame"] * F / (c["peak_tflops"] * 1e12) for c in mine)
                    tfl = fl * F / (per_step[name] * 1e-3) / 1e12
                    peak = fl * F / t_ideal / 1e12
                    rooflines[name] = dict(bound="mfma", achieved=tfl, peak=peak, unit="TFLOP/s (fp32-equivalent)",
                                           frac=tfl / peak, traffic=None, ms_per_step=per_step[name],
                                           algorithmic_flops_per_frame=fl, ms_at_peak=t_ideal * 1e3,
                                           frac_of_fp32_mfma_peak=tfl / PEAK_FP32_MFMA_TFLOPS, chains=mine)
            if per_step.get("sa_mlp", 0) > 0 and per_step.get("fp_mlp", 0) > 0:
                tms = per_step["sa_mlp"] + per_step["fp_mlp"]
                tfl = (sa_fl + fp_fl) * F / (tms * 1e-3) / 1e12
                t_ideal = sum(c["flops_per_frame"] * F / (c["peak_tflops"] * 1e12) for c in chains)
                peak = (sa_fl + fp_fl) * F / t_ideal / 1e12
                rooflines["sa_mlp+fp_mlp"] = dict(bound="mfma", achieved=tfl, peak=peak, unit="TFLOP/s (fp32-equivalent)",
                                                  frac=tfl / peak, traffic=None, ms_per_step=tms, ms_at_peak=t_ideal * 1e3,
                                                  algorithmic_flops_per_frame=sa_fl + fp_fl,
                                                  frac_of_fp32_mfma_peak=tfl / PEAK_FP32_MFMA_TFLOPS)
        if "ball_query" in op_step and "group" in op_step:
            per_step_bg = op_step
            tms = per_step_bg["ball_query"] + per_step_bg["group"]
            gbs = (alg["ball_query"] + alg["group"]) * F / (tms * 1e-3) / 1e9
            rooflines["ball_query+group"] = dict(bound="hbm", achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s",
                                                 frac=gbs / PEAK_HBM_GBS, traffic=None, ms_per_step=tms,
                                                 algorithmic_bytes_per_frame=alg["ball_query"] + alg["group"])
        if "vote_cluster_pose" in per_step:
            counts = res["counts"].cpu().numpy().astype(np.float64)
            # 16 flops per (seed, point) pair per iteration + the neighbour-count pass
            flops = 16.0 * float((iters.astype(np.float64) * counts * counts).sum())
            tfl = flops / (per_step["vote_cluster_pose"] * 1e-3) / 1e12
            rooflines["vote_cluster_pose"] = dict(bound="valu_fp32", achieved=tfl, peak=PEAK_FP32_VALU_TFLOPS,
                                                  unit="TFLOP/s", frac=tfl / PEAK_FP32_VALU_TFLOPS, traffic=None,
                                                  ms_per_step=per_step["vote_cluster_pose"],
                                                  algorithmic_flops_per_step=flops)
        # HBM traffic from the PMC passes of tools/pmc_traffic.py (same command, separate run)
        try:
            with open(os.path.join(ROOT, "profiles", "latest_pmc_traffic.json")) as f:
                pmc = json.load(f)
            if pmc.get("frames_per_step") == F and args.n_pts == 12288:
                sb = pmc["stage_bytes_per_step"]
                for name in rooflines:
                    parts = name.split("+")
                    if all(p in sb for p in parts):
                        rooflines[name]["traffic"] = sum(sb[p]["total_bytes"] for p in parts)
                        rooflines[name]["traffic_source"] = "profiles/%s_pmc_traffic.json" % pmc.get("tag")
                        # (PMC passes are separate rocprofv3 --pmc runs of the same command in another gpurun session, i.e.
                        # on another MI355X of the same pool: bytes per step are a property of the launches, not of the box)
                        rooflines[name]["traffic_measured_in_run"] = False
                        rooflines[name]["traffic_box"] = "separate gpurun session (another MI355X of the pool), same command"
        except (OSError, ValueError, KeyError):
            pass
        # dominant stage: the fused SA / FP MLP chains are ONE kernel family on the matrix pipe (their two event brackets are
        # taken together); everything else competes as measured
        leaf = {k: v for k, v in per_step.items() if k != "pointnet2_msg_total"}
        if "sa_mlp+fp_mlp" in rooflines:
            leaf = {k: v for k, v in leaf.items() if k not in ("sa_mlp", "fp_mlp")}
            leaf["sa_mlp+fp_mlp"] = per_step["sa_mlp"] + per_step["fp_mlp"]
        dominant = max(leaf, key=leaf.get) if leaf else None
        hm = {k: v for k, v in leaf.items() if rooflines.get(k, {}).get("bound") in ("hbm", "mfma")}
        dominant_hm = max(hm, key=hm.get) if hm else None
        if net is not None and "pointnet2_msg_total" in per_step:
            # torch glue inside the forward (transposes, concat, interpolation weights)
            per_step["pointnet2_msg_other"] = per_step["pointnet2_msg_total"] - sum(
                per_step.get(k, 0.0) for k in ("fps", "gather", "ball_query", "sa_mlp", "three_nn", "fp_mlp"))
        if net is None:
            island = ("Pointnet2MSG SA/FP op chain only (FPS, gather, ball_query, group, three_nn, "
                      "three_interpolate; synthetic features, no MLP GEMMs)")
        else:
            island = ("Pointnet2MSG forward (4 SA-MSG + 4 FP levels, random-init weights, eval): FPS, gather, "
                      "ball_query, fused group->SharedMLP->max-pool and three_nn, fused three_interpolate->SharedMLP; "
                      "fp32 operands and results throughout; the contraction runs as three exact fp16 x fp16 partial products "
                      "per multiply on fp16 MFMA with two fp16 pieces per operand (every SA level and FP levels 0-1 in one fused "
                      "kernel per chain -- SA levels 0-1 and FP level 0 the narrow-chain kernels with the chain's weights in LDS; "
                      "FP levels 2-3 and the pre-contractions layer by layer; every chain rescaled diagonally by powers of two "
                      "on the host (per-row weight scales), activations scaled from device-side bounds; PVN3D_MLP_ARITH=bf16x3 selects six bf16 x bf16 partial products with three bf16 pieces, fp32 the "
                      "fp32-MFMA kernels); all of them are as close "
                      "to an fp64 evaluation as the fp32 FMA chain (tests: 2e-5 of the output scale, measured 5e-7 - 1e-6)")
        out = {
            "metric": "frames/sec (12 288 pts, 8 kps) end-to-end vote+cluster+pose; idx bit-exact",
            "value": total_frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "LineMOD 'ape' eval path: %s + vote -> MeanShift x9 -> Kabsch; N=%d pts, "
                                   "n_obj=%d, K=8" % (island, args.n_pts, args.n_obj),
                       "frames_per_gpu_per_step": frames_local, "frames_per_step_all_gpus": frames_total,
                       "parallelism": "frames sharded x%d; one all-gather of per-frame result rows per step%s"
                                      % (world, "" if world > 1 else " (no-op at 1 GPU)"),
                       "streams": "1 (serial)" if args.serial else "3 (Pointnet2MSG feature path || its xyz-only geometry (FPS, ball query, three_nn) || "
                                  "vote-cluster-pose)" if net is not None else "2 (SA/FP ops || vote-cluster-pose)",
                       "pipelining": ("step i enqueues the xyz-only geometry of step i+1's batch beside its own MLP kernels "
                                      "(one geometry + one MLP + one vote pass per step inside the timed region; "
                                      "--no-geometry-ahead: every step alone, see value_self_contained_steps)") if geo_ahead
                       else "none (every step stands alone)",
                       "meanshift_kernel": args.ms_kernel,
                       # the arithmetic of the SA / FP contractions, as its own field (PVN3D_MLP_ARITH)
                       "arithmetic": mlp_arithmetic_field() if net is not None else None},
            "op_chain_stage_ms_per_step": op_step if net is not None else None,
            "stage_ms_per_step": per_step,
            "dominant_stage": dominant,
            # Two fixed keys.  `roofline` (the contract's: bound "hbm" | "mfma") = the largest stage that HAS such a roofline,
            # named in roofline.stage -- the fused SA / FP chain family on the matrix pipe.  `roofline_dominant_stage` = the
            # roofline of `dominant_stage` whatever bounds it: the vote stage is fp32-VALU / v_exp bound (16 flop per (seed,
            # point) pair), which is neither of the contract's two bounds; the same object is also under
            # `roofline_valu` (older readers) and in `rooflines`.
            "roofline": dict(rooflines.get(dominant_hm if dominant_hm in rooflines else "ball_query+group") or {},
                             stage=dominant_hm if dominant_hm in rooflines else "ball_query+group"),
            "roofline_dominant_stage": dict(rooflines.get(dominant) or {}, stage=dominant),
            "roofline_valu": rooflines.get("vote_cluster_pose"),
            "roofline_hbm": rooflines.get("ball_query+group"),
            "rooflines": rooflines,
            "meanshift_iters": {"min": int(iters.min()), "max": int(iters.max()), "mean": float(iters.mean())},
            "pose_err_vs_ground_truth": pose_err,
        }
        if elapsed_alone is not None:
            out["value_self_contained_steps"] = dict(value=total_frames / elapsed_alone, unit="frames/s",
                                                     ms_per_step=elapsed_alone / args.steps * 1e3,
                                                     note="every step with its own geometry at the head of its MLP stream "
                                                          "(no cross-step pipelining)")
        # the same vote -> cluster -> pose call with the reference's stop rule run to its end (no winner stop): the
        # iteration counts the reference would run on THESE frames, and that the poses are the same bits
        from pvn3d_amd.lib.utils import _vote_engine as _ve
        saved_k = _ve.DEFAULT_KERNEL
        _ve.DEFAULT_KERNEL = "nowin" if not saved_k else saved_k + "+nowin"
        try:
            res_full = run_postproc(inp, timer_off, args.poll_every)
        finally:
            _ve.DEFAULT_KERNEL = saved_k
        itf = res_full["iters"].cpu().numpy()
        out["meanshift_iters_reference_stop_rule"] = {"min": int(itf.min()), "max": int(itf.max()), "mean": float(itf.mean())}
        out["poses_identical_to_reference_stop_rule"] = bool(torch.equal(res["poses"], res_full["poses"])
                                                             and torch.equal(res["cls_kps"], res_full["cls_kps"]))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(f0)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            if "vote_cluster_pose" in per_step:
                # the north_star's ">= 10x the reference CPU MeanShift vote-clustering" on identical inputs:
                # one frame's CPU MeanShift time vs the GPU's vote->cluster->pose time per frame
                out["cpu_baseline"]["vote_clustering_gpu_over_cpu"] = (
                    out["cpu_baseline"]["seconds_per_frame"] * out["cpu_baseline"]["meanshift_share"]
                    / (per_step["vote_cluster_pose"] * 1e-3 / F))
        if world == 1 and not args.no_extra_configs and args.n_pts == 12288:
            out["device_copy"] = device_copy_bandwidth(dev)
            out["configs"] = extra_configs(net, dev, args.poll_every, not args.no_cpu_baseline)
            # the headline converges in ~4 MeanShift iterations per fit; the heavy-tailed vote set (47-274
            # iterations, the range SURVEY.md section 6 saw on the reference) is the hard case: its own top-level value
            for c in out["configs"]:
                if c.get("name") == "heavy_tail_votes":
                    out["value_heavy_tail"] = dict(value=c["frames_per_s"], unit="frames/s (vote -> cluster -> pose only)",
                                                   ms_per_frame=c["ms_per_frame"], meanshift_iters=c["meanshift_iters"])
        if train_dist is not None:
            out["configs"] = [train_dist]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
