"""Pointnet2MSG -- the PointNet++ branch of PVN3D with the reference's hyper-parameters and
module tree (pvn3d/lib/pvn3d.py:46-154: 4 multi-scale set-abstraction levels, 4 feature
propagation levels), built from this package's SA/FP modules.  Only this class of
lib/pvn3d.py is on the hot path; the CNN, DenseFusion and heads are out of scope."""
import os

import torch
import torch.nn as nn

from .pointnet2_utils import pointnet2_modules as _pm
from .pointnet2_utils.pointnet2_modules import PointnetSAModuleMSG, PointnetFPModule

# Inference: FPS / ball query / three_nn of every level depend on xyz only.  They are
# latency-bound (one workgroup per frame for FPS), so they run ahead on a second HIP stream
# while the MFMA kernels of the previous level occupy the matrix cores; the feature path waits
# on one event per level.  False keeps everything on the caller's stream.
GEOMETRY_STREAM = os.environ.get("PVN3D_GEOMETRY_STREAM", "1") != "0"
# The fp16 x 2 chains' input bounds (abs-max of the cloud and of its features) with the geometry handle, on the geometry
# stream; "0" computes them at the head of the feature path as rounds 5 did (A/B)
BOUNDS_AHEAD = os.environ.get("PVN3D_BOUNDS_AHEAD", "1") != "0"
_geo_streams = {}


def _geo_stream(dev):
    st = _geo_streams.get(dev)
    if st is None:
        st = _geo_streams[dev] = torch.cuda.Stream(device=dev)
    return st


class GraphedForward(object):
    """A HIP-graph replay of ``net(pointcloud)`` for ONE static input shape (eval mode, no autograd): the ~70
    launches of the fused forward -- both streams of it, the geometry stream joins the capture through its events
    -- become one graph launch.  The reference evaluates one frame per call (test_mini_batch_size = 1,
    pvn3d/common.py:41): at B = 1 the forward is launch-bound outside FPS.
    ``g = GraphedForward(net, example); out = g(pc)``: `pc` is copied into the captured input buffer, the returned
    tensor is the captured output buffer (overwritten by the next call).
    The graph bakes in the folded Conv+BatchNorm weights of the capture: a later ``load_state_dict`` / optimizer or
    training step / ``broadcast_parameters`` is detected (tensor versions + the package's weights epoch) and the
    graph is captured again before the replay; ``check_weights=False`` skips that check (~15 us per call)."""

    def __init__(self, net, example, warmup=3, check_weights=True):
        assert not net.training and example.is_cuda
        self.net = net
        self.check_weights = check_weights
        self._warmup = warmup
        self._capture(example)

    def _signature(self):
        from .pointnet2_utils import _fused_mlp
        sig = [_fused_mlp._WEIGHTS_EPOCH[0]]
        for t in self.net.parameters():
            sig.append(t._version)
        for t in self.net.buffers():
            sig.append(t._version)
        return tuple(sig)

    def _capture(self, example):
        net, warmup = self.net, self._warmup
        self._sig = self._signature()
        self.static_in = example.clone()
        cur = torch.cuda.current_stream(example.device)
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():      # warm-up off the capture: one-time function attributes,
            for _ in range(warmup):                          # allocator pools, packed weights
                net(self.static_in)
        cur.wait_stream(side)
        torch.cuda.synchronize(example.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = net(self.static_in)

    def __call__(self, pointcloud):
        if pointcloud.shape != self.static_in.shape:
            raise RuntimeError("GraphedForward was captured for shape %s" % (tuple(self.static_in.shape),))
        if self.check_weights and self._signature() != self._sig:
            if self.net.training:
                raise RuntimeError("GraphedForward replays the eval forward: the network is in training mode")
            self._capture(self.static_in)          # the weights changed since the capture
        self.static_in.copy_(pointcloud)
        self.graph.replay()
        return self.static_out


class Pointnet2MSG(nn.Module):
    def __init__(self, input_channels=6, use_xyz=True):
        super(Pointnet2MSG, self).__init__()
        self.SA_modules = nn.ModuleList()
        c_in = input_channels
        self.SA_modules.append(PointnetSAModuleMSG(
            npoint=2048, radii=[0.0175, 0.025], nsamples=[16, 32],
            mlps=[[c_in, 16, 16, 32], [c_in, 32, 32, 64]], use_xyz=use_xyz))
        c_out_0 = 32 + 64
        self.SA_modules.append(PointnetSAModuleMSG(
            npoint=1024, radii=[0.025, 0.05], nsamples=[16, 32],
            mlps=[[c_out_0, 64, 64, 128], [c_out_0, 64, 96, 128]], use_xyz=use_xyz))
        c_out_1 = 128 + 128
        self.SA_modules.append(PointnetSAModuleMSG(
            npoint=512, radii=[0.05, 0.1], nsamples=[16, 32],
            mlps=[[c_out_1, 128, 196, 256], [c_out_1, 128, 196, 256]], use_xyz=use_xyz))
        c_out_2 = 256 + 256
        self.SA_modules.append(PointnetSAModuleMSG(
            npoint=128, radii=[0.1, 0.2], nsamples=[16, 32],
            mlps=[[c_out_2, 256, 256, 512], [c_out_2, 256, 384, 512]], use_xyz=use_xyz))
        c_out_3 = 512 + 512
        self.FP_modules = nn.ModuleList()
        self.FP_modules.append(PointnetFPModule(mlp=[256 + input_channels, 128, 128]))
        self.FP_modules.append(PointnetFPModule(mlp=[512 + c_out_0, 256, 256]))
        self.FP_modules.append(PointnetFPModule(mlp=[512 + c_out_1, 512, 512]))
        self.FP_modules.append(PointnetFPModule(mlp=[c_out_3 + c_out_2, 512, 512]))

        # inference: the SA levels and the intermediate FP levels hand point-major buffers (as transposed
        # views) to the next level; the last one (FP_modules[0]) returns the reference's contiguous (B, 128, N)
        for fp in list(self.FP_modules)[1:]:
            fp._point_major_out = True
        for sa in self.SA_modules:
            sa._point_major_out = True

    def graphed(self, example):
        """-> GraphedForward(self, example): HIP-graph replay of the eval forward for example's shape."""
        return GraphedForward(self, example)

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    def sampling_ahead(self, pointcloud, stream=None):
        """The first level's FPS run of `pointcloud` alone, on `stream` (default: the geometry stream), ordered after
        the current stream: -> (sel, dmax, event) for ``geometry_ahead(pointcloud, presampled=...)``.  A pipelined
        evaluator runs it one batch further ahead than the rest of the geometry (lib/pipeline.py, depth 3)."""
        xyz = pointcloud[..., 0:3].contiguous()
        cur = torch.cuda.current_stream(xyz.device)
        st = stream if stream is not None else _geo_stream(xyz.device)
        xyz.record_stream(st)
        st.wait_stream(cur)
        plan = [m.npoint for m in list(self.SA_modules)[1:]]
        with torch.cuda.stream(st):
            sel, dmax = self.SA_modules[0].sample_first_level(xyz, plan=plan)
            ev = torch.cuda.Event()
            ev.record(st)
        for t in (sel, dmax):
            if t is not None:
                t.record_stream(cur)
        return sel, dmax, ev

    def geometry_ahead(self, pointcloud, presampled=None):
        """Software pipelining across batches: enqueue the xyz-only work (FPS, ball query, three_nn of
        every level) of `pointcloud` on the geometry stream NOW and return a handle for a later
        ``forward(pointcloud, geometry=handle)``.  An evaluator calls this for batch s+1 before it runs
        ``forward`` on batch s: the latency-bound FPS chain (one workgroup per cloud, ~4 ms, 64 of 256 CUs)
        then runs beside the MFMA kernels of batch s instead of in front of those of batch s+1.  The work is
        ordered after everything already enqueued on the current stream (the producer of `pointcloud`).
        (Measured on the 64-frame bench: no throughput gain -- the GPU is already busy with the other island's
        kernels while FPS runs -- so bench.py does not use it; it shortens the latency of a stream of small
        batches.)"""
        xyz = pointcloud[..., 0:3].contiguous()
        # `xyz` is allocated on the current stream but read by kernels of the geometry stream after this
        # function has returned and dropped its reference: keep the allocator from recycling it early
        xyz.record_stream(_geo_stream(xyz.device))
        sa_geo, fp_geo = self._geometry_ahead(xyz, presampled=presampled)
        h = {"sa": sa_geo, "fp": fp_geo, "shape": tuple(xyz.shape), "device": xyz.device, "xyz": xyz}
        h.update(self._input_bounds(pointcloud, xyz))
        return h

    def _input_bounds(self, pointcloud, xyz):
        """The device-side abs-max bounds of the cloud's coordinates and input features that the fp16 x 2 chains scale
        their operands by -- input-only work like the geometry, so it runs on the geometry stream too (round 6: the two
        reductions were 39 us at the head of the feature path, the step's critical stream).  They are enqueued behind the
        levels' geometry, so the feature path waits for `bounds_event`, not for a level's hand-over event."""
        from .pointnet2_utils import _ext, _fused_mlp
        if _fused_mlp.MLP_ARITH != "fp16x2" or not BOUNDS_AHEAD:
            return {}
        cur = torch.cuda.current_stream(xyz.device)
        geo = _geo_stream(xyz.device)
        pointcloud.record_stream(geo)
        with torch.cuda.stream(geo):
            xb = _ext.table_absmax(xyz, xyz.size(0) * xyz.size(1), 3, 3)
            fb = None
            c = pointcloud.size(-1) - 3
            if c > 0:
                feats = pointcloud[..., 3:].transpose(1, 2)        # the view the feature path reads in place
                fb = _ext.table_absmax(feats, xyz.size(0) * xyz.size(1), c, pointcloud.size(-1))
            ev = torch.cuda.Event()
            ev.record(geo)
        for t in (xb, fb):
            if t is not None:
                t.record_stream(cur)
        return {"xyz_bound": xb, "feat_bound": fb, "bounds_event": ev}

    def _geometry_ahead(self, xyz, presampled=None):
        """Run every level's xyz-only work on the geometry stream; returns per-level results and
        the events the feature path has to wait for."""
        cur = torch.cuda.current_stream(xyz.device)
        geo = _geo_stream(xyz.device)
        geo.wait_stream(cur)
        sa_geo, fp_geo, l_xyz = [], [None] * len(self.FP_modules), [xyz]

        def hand_over(tensors):
            ev = torch.cuda.Event()
            ev.record(geo)
            for t in tensors:
                if t is not None:
                    t.record_stream(cur)      # allocated on `geo`, consumed on `cur`
            return ev

        pre = None
        if presampled is not None:
            sel, dmax, pre_ev = presampled       # `pointcloud`'s first-level FPS run (sampling_ahead), made earlier
            if pre_ev is not None:
                geo.wait_event(pre_ev)
            for t in (sel, dmax):
                if t is not None:
                    t.record_stream(geo)
            pre = (sel, dmax)
        with torch.cuda.stream(geo):
            nest = None
            for li, sa in enumerate(self.SA_modules):
                # every level samples the previous level's centres in the order they were picked: one FPS run
                # (the first level's) plus one verification pass decide the whole pyramid
                plan = [m.npoint for m in list(self.SA_modules)[1:]] if li == 0 else None
                new_xyz, idxs, nest = sa.sample_and_query_nested(l_xyz[-1], nest=nest, plan=plan,
                                                                 presampled=pre if li == 0 else None)
                sa_geo.append(((new_xyz, idxs), hand_over([new_xyz] + list(idxs))))
                l_xyz.append(new_xyz)
            for i in range(-1, -(len(self.FP_modules) + 1), -1):
                idx, weight = PointnetFPModule.neighbours(l_xyz[i - 1], l_xyz[i])
                fp_geo[i] = ((idx, weight), hand_over([idx, weight]))
        return sa_geo, fp_geo

    def forward(self, pointcloud, geometry=None):
        """pointcloud (B, N, 3 + input_channels) -> per-point features (B, 128, N).
        geometry: optional handle from ``geometry_ahead(pointcloud)``: the xyz-only work (FPS, ball query, three_nn) of
        THIS cloud, enqueued earlier on the geometry stream -- by an evaluator for the next batch, or by a training loop
        for the next step's batch while the current step's backward runs (train_step.py's ``prefetch``).  The indices
        are what an inline run would compute (no parameters are involved), so training results do not change."""
        in_place = (pointcloud.size(-1) > 3 and not self.training and _pm.FUSED_INFERENCE and not torch.is_grad_enabled())
        if in_place:
            # the fused levels read the features in place (point-major view: same values) -- the channel-major copy of
            # _break_up_pc would be 19 MB written and dropped per 64-frame forward; the contiguous xyz of a geometry
            # handle made for THIS cloud is reused instead of copied again
            xyz = geometry["xyz"] if geometry is not None and geometry.get("xyz") is not None \
                else pointcloud[..., 0:3].contiguous()
            features = pointcloud[..., 3:].transpose(1, 2)
            if xyz.is_cuda:
                xyz.record_stream(torch.cuda.current_stream(xyz.device))      # (a handle's xyz may come from another stream)
        else:
            xyz, features = self._break_up_pc(pointcloud)
        ahead = (GEOMETRY_STREAM and _pm.FUSED_INFERENCE and not self.training and xyz.is_cuda
                 and not torch.is_grad_enabled())
        l_xyz, l_features = [xyz], [features]
        if geometry is not None and not ahead and not (self.training and xyz.is_cuda and GEOMETRY_STREAM):
            raise RuntimeError("a geometry handle is used by the eval fast path (GEOMETRY_STREAM, FUSED_INFERENCE, no "
                               "autograd) and by training on CUDA: drop it or change the mode")
        if geometry is not None and not ahead:
            ahead = True                   # training with a handle prepared ahead (train_step's prefetch)
        if not ahead:
            nest = None
            for li, sa in enumerate(self.SA_modules):
                plan = [m.npoint for m in list(self.SA_modules)[1:]] if li == 0 else None
                new_xyz, idxs, nest = sa.sample_and_query_nested(l_xyz[-1], nest=nest, plan=plan)
                li_xyz, li_features = sa(l_xyz[-1], l_features[-1], geometry=(new_xyz, idxs))
                l_xyz.append(li_xyz)
                l_features.append(li_features)
            for i in range(-1, -(len(self.FP_modules) + 1), -1):
                l_features[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_features[i - 1], l_features[i])
            return l_features[0]
        from .pointnet2_utils import _ext as _ext_z
        with _ext_z.zero_arena(xyz.device):            # the levels' abs-max words and bounds: one fill instead of a dozen
            return self._forward_ahead(pointcloud, xyz, features, geometry, l_xyz, l_features)

    def _forward_ahead(self, pointcloud, xyz, features, geometry, l_xyz, l_features):
        cur = torch.cuda.current_stream(xyz.device)
        if geometry is not None:
            if geometry["shape"] != tuple(xyz.shape) or geometry["device"] != xyz.device:
                raise RuntimeError("geometry handle was built for a cloud of shape %s on %s, not %s on %s"
                                   % (geometry["shape"], geometry["device"], tuple(xyz.shape), xyz.device))
            sa_geo, fp_geo = geometry["sa"], geometry["fp"]
            for (geom, _), in [(g,) for g in sa_geo]:
                for t in [geom[0]] + list(geom[1]):
                    if t is not None:
                        t.record_stream(cur)       # produced on the geometry stream, consumed on THIS stream
        else:
            sa_geo, fp_geo = self._geometry_ahead(xyz)
        # fp16 x 2 chains scale their operands by device-side bounds: one abs-max of the cloud bounds every level's
        # centres (they are points of the cloud), so the levels inherit it instead of reducing their own xyz
        xyz_bound = None
        from .pointnet2_utils import _ext, _fused_mlp
        if _fused_mlp.MLP_ARITH == "fp16x2":
            if geometry is not None and geometry.get("xyz_bound") is not None:
                # reduced on the geometry stream with the handle (input-only work)
                cur.wait_event(geometry["bounds_event"])
                xyz_bound = geometry["xyz_bound"]
                if features is not None and geometry.get("feat_bound") is not None:
                    features._pvn3d_bound = geometry["feat_bound"]
            else:
                xyz_bound = _ext.table_absmax(xyz, xyz.size(0) * xyz.size(1), 3, 3)
        for sa, (geom, ev) in zip(self.SA_modules, sa_geo):
            cur.wait_event(ev)
            if xyz_bound is not None:
                geom[0]._pvn3d_bound = xyz_bound
            li_xyz, li_features = sa(l_xyz[-1], l_features[-1], geometry=geom)
            l_xyz.append(li_xyz)
            l_features.append(li_features)
        for i in range(-1, -(len(self.FP_modules) + 1), -1):
            nbrs, ev = fp_geo[i]
            cur.wait_event(ev)
            l_features[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_features[i - 1], l_features[i],
                                                   neighbours=nbrs)
        return l_features[0]
