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

Extra JSON objects: ``roofline`` (dominant kernel, measured with events inside the timed region),
``rooflines`` (every stage), ``cpu_baseline`` (the reference's dense-torch MeanShift + numpy
Kabsch restated in oracle/, timed on the host cores on a bounded sample; rank 0, N=1 only).
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
        g0 = _ext.group_xyz_features(xyz, new_xyz, feats, i0, True)
        g1 = _ext.group_xyz_features(xyz, new_xyz, feats, i1, True)
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


def run_net(net, inp, timer):
    """(A) the fused Pointnet2MSG forward; `timer` (if enabled) brackets its stages."""
    from pvn3d_amd.lib.pointnet2_utils import pointnet2_modules as pm
    from pvn3d_amd.lib import pointnet2_msg
    pm.STAGE_HOOK = (lambda name: _HookStage(timer, name)) if timer.enabled else None
    ahead = pointnet2_msg.GEOMETRY_STREAM
    if timer.enabled:
        pointnet2_msg.GEOMETRY_STREAM = False     # stage events need one serial stream
    try:
        with torch.no_grad():
            t = timer.start("pointnet2_msg_total")
            out = net(inp["pc"])
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


def run_postproc(inp, timer, poll_every):
    """(B) vote -> MeanShift x (K+1) -> Kabsch for the whole batch."""
    from pvn3d_amd.lib.utils import pvn3d_eval_utils as ev
    t = timer.start("vote_cluster_pose")
    res = ev.cal_batch_poses_lm(inp["pcld"], inp["mask"], inp["ctr_of"].unsqueeze(1) if inp["ctr_of"].dim() == 3
                                else inp["ctr_of"], inp["pred_kp_of"], True, 2, False, 1, poll_every=poll_every)
    timer.stop(t)
    return res


def cpu_baseline(frame, budget_s=25.0):
    """Reference CPU path (dense torch MeanShift + numpy Kabsch, oracle/torch_port.py) on ONE
    frame of the same workload on the host cores; plus the C/OpenMP oracle for comparison.
    Bounded: if the first fit predicts more than `budget_s` for the frame, only the centre fit
    and the first keypoint fit are run and the 9-fit frame time is extrapolated (and labelled)."""
    from oracle import posecal, torch_port, native
    cores = os.cpu_count() or 1
    threads = min(cores, 32)          # the dense (n,n,3) temporaries are memory-bound
    torch.set_num_threads(threads)
    n_fit = [0]
    t_fit = []

    class _Stop(Exception):
        pass

    def fit(A, bw):
        t0 = time.perf_counter()
        c, l, it = torch_port.meanshift_fit_dense(torch.from_numpy(np.ascontiguousarray(A)), bw)
        t_fit.append(time.perf_counter() - t0)
        n_fit[0] += 1
        if n_fit[0] >= 2 and sum(t_fit) / len(t_fit) * 9 > budget_s:
            raise _Stop()
        return c.numpy(), l.numpy(), it
    extrapolated = False
    t0 = time.perf_counter()
    try:
        posecal.cal_frame_poses_lm(frame["pcld"], frame["mask"], frame["ctr_of"], frame["pred_kp_of"], True, 2,
                                   False, frame["mesh_kps"], fit=fit, bft=torch_port.best_fit_transform_np)
        t_ref = time.perf_counter() - t0
    except _Stop:
        extrapolated = True
        t_ref = sum(t_fit) / len(t_fit) * 9
    native.set_num_threads(min(cores, 64))
    t0 = time.perf_counter()
    posecal.cal_frame_poses_lm(frame["pcld"], frame["mask"], frame["ctr_of"], frame["pred_kp_of"], True, 2, False,
                               frame["mesh_kps"])
    t_c = time.perf_counter() - t0
    n_obj = int((frame["mask"] == 1).sum())
    sample = ("1 frame (N=%d, n_obj=%d, 9 fits) vote+cluster+pose, dense torch-CPU restatement of "
              "MeanShiftTorch.fit + numpy Kabsch" % (len(frame["pcld"]), n_obj))
    if extrapolated:
        sample += "; %d of 9 fits timed (%.1f s), frame time extrapolated x9/%d" % (len(t_fit), sum(t_fit), len(t_fit))
    return dict(value=1.0 / t_ref, unit="frames/s", cores=threads, kind="port", sample=sample,
                seconds_per_frame=t_ref, host_cores_available=cores,
                c_openmp_port=dict(value=1.0 / t_c, unit="frames/s", cores=min(cores, 64),
                                   seconds_per_frame=t_c))


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
    ap.add_argument("--ops-only", action="store_true",
                    help="island (A) = bare SA/FP op chain with synthetic features (no MLP GEMMs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"

    scale = args.n_pts / 12288.0
    inp = make_inputs(args.frames, args.n_pts, args.n_obj, dev, seed_base=1000 * rank)
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
    side = torch.cuda.Stream(device=dev)

    def step(timer):
        if args.serial:
            keep = island_a(timer)
            res = run_postproc(inp, timer, args.poll_every)
            return keep, res
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            keep = island_a(timer_off)
        res = run_postproc(inp, timer_off, args.poll_every)
        torch.cuda.current_stream(dev).wait_stream(side)
        return keep, res

    for _ in range(args.warmup):
        step(timer_off)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer = StageTimer(args.serial and not args.no_stage_events)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        keep, res = step(timer)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
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
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the timed result is the real thing (pose close to the synthetic ground truth)
    pose0 = res["poses"][0].cpu().numpy()
    f0 = inp["frames"][0]
    pose_err = float(max(np.abs(pose0[:, :3] - f0["R"]).max(), np.abs(pose0[:, 3] - f0["t"]).max()))
    iters = res["iters"].cpu().numpy()

    if rank == 0:
        total_frames = args.frames * world * args.steps
        stage_ms = timer.totals_ms()
        per_step = {k: v / args.steps for k, v in stage_ms.items()}
        op_step = {k: v / args.steps for k, v in op_timer.totals_ms().items()}
        alg = algorithmic_bytes_per_frame(args.n_pts)
        rooflines = {}
        F = args.frames
        for name in ["ball_query", "group", "three_interpolate", "three_nn", "gather", "fps"]:
            if name in op_step and op_step[name] > 0:
                gbs = alg[name] * F / (op_step[name] * 1e-3) / 1e9
                rooflines[name] = dict(bound="hbm", achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s",
                                       frac=gbs / PEAK_HBM_GBS, traffic=None, ms_per_step=op_step[name],
                                       algorithmic_bytes_per_frame=alg[name])
        if net is not None:
            sa_fl, fp_fl = mlp_flops_per_frame(net, scale)
            for name, fl in (("sa_mlp", sa_fl), ("fp_mlp", fp_fl)):
                if per_step.get(name, 0) > 0:
                    tfl = fl * F / (per_step[name] * 1e-3) / 1e12
                    rooflines[name] = dict(bound="mfma", achieved=tfl, peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                                           frac=tfl / PEAK_FP32_MFMA_TFLOPS, traffic=None,
                                           ms_per_step=per_step[name], algorithmic_flops_per_frame=fl)
            if per_step.get("sa_mlp", 0) > 0 and per_step.get("fp_mlp", 0) > 0:
                tms = per_step["sa_mlp"] + per_step["fp_mlp"]
                tfl = (sa_fl + fp_fl) * F / (tms * 1e-3) / 1e12
                rooflines["sa_mlp+fp_mlp"] = dict(bound="mfma", achieved=tfl, peak=PEAK_FP32_MFMA_TFLOPS,
                                                  unit="TFLOP/s", frac=tfl / PEAK_FP32_MFMA_TFLOPS, traffic=None,
                                                  ms_per_step=tms, algorithmic_flops_per_frame=sa_fl + fp_fl)
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
        except (OSError, ValueError, KeyError):
            pass
        leaf = {k: v for k, v in per_step.items() if k != "pointnet2_msg_total"}
        dominant = max(leaf, key=leaf.get) if leaf else None
        if net is not None and "pointnet2_msg_total" in per_step:
            # torch glue inside the forward (transposes, concat, interpolation weights)
            per_step["pointnet2_msg_other"] = per_step["pointnet2_msg_total"] - sum(
                per_step.get(k, 0.0) for k in ("fps", "gather", "ball_query", "sa_mlp", "three_nn", "fp_mlp"))
        if net is None:
            island = ("Pointnet2MSG SA/FP op chain only (FPS, gather, ball_query, group, three_nn, "
                      "three_interpolate; synthetic features, no MLP GEMMs)")
        else:
            island = ("Pointnet2MSG forward (4 SA-MSG + 4 FP levels, random-init weights, eval): FPS, gather, "
                      "ball_query, fused group->SharedMLP->max-pool and three_nn, fused three_interpolate->SharedMLP "
                      "on fp32 MFMA")
        out = {
            "metric": "frames/sec (12 288 pts, 8 kps) end-to-end vote+cluster+pose; idx bit-exact",
            "value": total_frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "LineMOD 'ape' eval path: %s + vote -> MeanShift x9 -> Kabsch; N=%d pts, "
                                   "n_obj=%d, K=8" % (island, args.n_pts, args.n_obj),
                       "frames_per_gpu_per_step": args.frames, "parallelism": "frames sharded x%d, no collective" % world,
                       "streams": "1 (serial)" if args.serial else "3 (Pointnet2MSG feature path || its xyz-only geometry (FPS, ball query, three_nn) || "
                                  "vote-cluster-pose)" if net is not None else "2 (SA/FP ops || vote-cluster-pose)"},
            "op_chain_stage_ms_per_step": op_step if net is not None else None,
            "stage_ms_per_step": per_step,
            "dominant_stage": dominant,
            "roofline": rooflines.get(dominant if dominant in rooflines else "ball_query+group"),
            "roofline_hbm": rooflines.get("ball_query+group"),
            "rooflines": rooflines,
            "meanshift_iters": {"min": int(iters.min()), "max": int(iters.max()), "mean": float(iters.mean())},
            "pose_err_vs_ground_truth": pose_err,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(f0)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
