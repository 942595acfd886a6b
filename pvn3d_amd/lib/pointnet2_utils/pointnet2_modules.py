"""Set-abstraction / feature-propagation modules with the reference's constructor signatures,
attribute names (``npoint``, ``groupers``, ``mlps``, ``mlp``) and ``state_dict`` layout
(pvn3d/lib/pointnet2_utils/pointnet2_modules.py: _PointnetSAModuleBase :20-71,
PointnetSAModuleMSG :74-112, PointnetSAModule :115-143, PointnetFPModule :146-206).

Data flow per SA level (reference :47-71):
    FPS -> gather centres -> per scale: ball_query -> group(xyz-rel ++ features) -> SharedMLP
    -> max over nsample -> concat scales.
Here the two ball queries of a multi-scale level share one scan of the cloud
(_ext.ball_query_pair) and grouping writes the concatenated tensor in one pass.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _ext
from . import _fused_mlp
from . import _train_mlp
from . import _small_batch
from . import pointnet2_utils
from ..utils import pytorch_utils as pt_utils

# Inference fast path: gather -> SharedMLP -> max-pool (and interpolate -> SharedMLP) as one
# fp32-MFMA kernel each (csrc/sa_mlp.hip).  Used when the module is in eval mode and autograd
# is not recording; set to False to force the reference's op-by-op composition.
FUSED_INFERENCE = True
# Pyramid levels after the first take their FPS result from the first level's run when it is provably the
# same (sample_and_query_nested); False = every level runs its own FPS.
FPS_NESTING = True


# Optional measurement hook: a callable ``name -> context manager`` bracketing each stage of the
# forward ("fps", "gather", "ball_query", "sa_mlp", "three_nn", "fp_mlp"); bench.py installs an
# event-pair timer here.  None (the default) costs one attribute read per stage.
STAGE_HOOK = None


class _NullStage(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_STAGE = _NullStage()


def _stage(name):
    hook = STAGE_HOOK
    return hook(name) if hook is not None else _NULL_STAGE


def _no_grad_needed(*tensors):
    if not torch.is_grad_enabled():
        return True
    return not any(t is not None and t.requires_grad for t in tensors)



class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super(_PointnetSAModuleBase, self).__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def _shared_idx(self, xyz, new_xyz):
        """Precompute neighbour indices when two ball-query scales can share one scan."""
        g = self.groupers
        if (len(g) == 2 and all(isinstance(m, pointnet2_utils.QueryAndGroup) for m in g)
                and xyz.is_cuda):
            return _ext.ball_query_pair(new_xyz, xyz, g[0].radius, g[0].nsample,
                                        g[1].radius, g[1].nsample)
        return [None] * len(g)

    def sample_and_query(self, xyz):
        """The geometry half of forward(): FPS centres and the neighbour indices of every scale.
        Depends on xyz only, so a caller may run it ahead of the feature path (another stream)
        and hand the result back through forward(..., geometry=...)."""
        if self.npoint is None:
            return None, [None] * len(self.groupers)
        with _stage("fps"):
            sel = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        with _stage("gather"):
            xyz_t = xyz.transpose(1, 2).contiguous()
            new_xyz = pointnet2_utils.gather_operation(xyz_t, sel).transpose(1, 2).contiguous()
        with _stage("ball_query"):
            idxs = self._shared_idx(xyz, new_xyz)
        return new_xyz, idxs

    @staticmethod
    def _nest_follow(npoint, plan):
        follow = []
        for m in (plan or [])[:3]:
            if m is None or m > (follow[-1] if follow else npoint):
                break
            follow.append(int(m))
        if follow and max(follow) > 8192:
            follow = []        # fps_nest_verify stages at most 8192 picks in LDS: every level runs its own FPS
        return follow

    def sample_first_level(self, xyz, plan=None):
        """The FPS run of a pyramid's first level alone: -> (sel, dmax or None), the `presampled` argument of
        sample_and_query_nested.  (lib/pipeline.py runs it one batch further ahead than the rest of the geometry: it is
        the one latency-bound kernel of the chain, 1.5 of its 2.1 ms.)"""
        with _stage("fps"):
            return _ext.furthest_point_sampling_nested(xyz, self.npoint, want_dmax=bool(self._nest_follow(self.npoint, plan)))

    def sample_and_query_nested(self, xyz, nest=None, plan=None, presampled=None):
        """sample_and_query for the levels of a pyramid whose next level samples THIS level's centres in the
        order they were picked (Pointnet2MSG): FPS is greedy, so the next level's run is the identity prefix
        unless a tie breaks differently -- checked once, for up to three following levels, by
        _ext.fps_nest_verify right after the first level (csrc/sampling.hip).  `plan` (first level): npoint of
        the following levels; `nest` (later levels): the state the previous level returned.
        -> (new_xyz, idxs, state for the next level or None).  Index-exact with sample_and_query."""
        if self.npoint is None or not (FPS_NESTING and xyz.is_cuda):
            return self.sample_and_query(xyz) + (None,)
        state = None
        with _stage("fps"):
            if nest is not None:
                flags, level = nest
                sel, _ = _ext.furthest_point_sampling_nested(xyz, self.npoint, nest=(flags, level))
                state = (flags, level + 1) if level + 1 < 3 else None
            else:
                follow = self._nest_follow(self.npoint, plan)
                if presampled is not None:
                    sel, dmax = presampled       # (first level only) this cloud's FPS run, made earlier
                else:
                    sel, dmax = _ext.furthest_point_sampling_nested(xyz, self.npoint, want_dmax=bool(follow))
        with _stage("gather"):
            xyz_t = xyz.transpose(1, 2).contiguous()
            new_xyz = pointnet2_utils.gather_operation(xyz_t, sel).transpose(1, 2).contiguous()
        if nest is None and follow:
            with _stage("fps"):
                state = (_ext.fps_nest_verify(new_xyz, dmax, follow), 0)
        with _stage("ball_query"):
            idxs = self._shared_idx(xyz, new_xyz)
        return new_xyz, idxs, state

    def forward(self, xyz, features=None, geometry=None):
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), new_features (B,sum(mlp[-1]),npoint)"""
        new_xyz, idxs = geometry if geometry is not None else self.sample_and_query(xyz)

        pooled = []
        fuse = (FUSED_INFERENCE and not self.training and self.npoint is not None and xyz.is_cuda
                and _no_grad_needed(xyz, features)
                and not any(p.requires_grad and torch.is_grad_enabled() for p in self.parameters()))
        if fuse:
            out = self._forward_fused(xyz, new_xyz, features, idxs)
            if out is not None:
                return new_xyz, out
        if (_train_mlp.train_fused_enabled() and self.training and self.npoint is not None and xyz.is_cuda
                and torch.is_grad_enabled() and (features is None or features.dtype == torch.float32)):
            # training: gather -> bf16 MFMA GEMM + BatchNorm(batch statistics) + ReLU chain -> max-pool, forward and
            # backward on the kernels of csrc/mlp_train.hip
            tidx = [idx if idx is not None else pointnet2_utils.ball_query(g.radius, g.nsample, xyz, new_xyz)
                    for g, idx in zip(self.groupers, idxs)] \
                if all(isinstance(g, pointnet2_utils.QueryAndGroup) for g in self.groupers) else None
            out = _train_mlp.sa_level_train(self, xyz, new_xyz, features, tidx) if tidx is not None else None
            if out is not None:
                return new_xyz, out
        if features is not None and not features.is_contiguous():
            features = features.contiguous()       # a point-major view from a fused producer
        for grouper, mlp, idx in zip(self.groupers, self.mlps, idxs):
            if idx is not None:
                grouped = grouper(xyz, new_xyz, features, idx=idx)
            else:
                grouped = grouper(xyz, new_xyz, features)      # (B, C, npoint, nsample)
            feats = mlp(grouped)                               # (B, mlp[-1], npoint, nsample)
            feats = F.max_pool2d(feats, kernel_size=[1, feats.size(3)]).squeeze(-1)
            pooled.append(feats)
        return new_xyz, torch.cat(pooled, dim=1)

    def _forward_fused(self, xyz, new_xyz, features, idxs):
        """Every scale as one fp32-MFMA kernel writing its slice of ONE point-major
        (B, npoint, sum(mlp[-1])) buffer; returns (B, C_out, npoint) with the values of
        torch.cat(pooled, 1), or None when a scale cannot be fused."""
        packs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            if not isinstance(grouper, pointnet2_utils.QueryAndGroup):
                return None
            ns = grouper.nsample
            if (ns & (ns - 1)) != 0 or ns > 64:
                return None
            packed = _fused_mlp.pack_shared_mlp(mlp, n_xyz_first=3 if (grouper.use_xyz and features is not None) else 0)
            if packed is None:
                return None
            packs.append(packed)
        widths = [p.dims[-1] for p in packs]
        offs = [0]
        for w in widths:
            offs.append(offs[-1] + (w + 3) // 4 * 4)      # 16-byte aligned slices
        if any((w & 3) for w in widths[:-1]):
            return None                                     # slices would not be contiguous channels
        total = offs[-1]
        out_pm = torch.empty((xyz.size(0), new_xyz.size(1), total), dtype=torch.float32, device=xyz.device)
        # first conv's feature half per source point, ahead of the gather (wide levels, full batches: _ext.sa_precontract)
        pre = None
        cols_min = min(xyz.size(0) * new_xyz.size(1) * g.nsample for g in self.groupers)
        if features is not None and all(g.use_xyz for g in self.groupers) and cols_min >= 64 * _small_batch.MAX_FUSED_WGS:
            with _stage("sa_mlp"):
                pre = _ext.sa_precontract(features, packs, [g.nsample for g in self.groupers])
        # fp16 x 2 chains leave the abs-max of what they write for the level that consumes this table (no extra pass);
        # it only counts if EVERY scale of the level went through such a chain
        amax = _ext.zeros_f32(1, xyz.device) if _fused_mlp.MLP_ARITH == "fp16x2" else None
        amax_writers = 0
        for si, (grouper, mlp, packed, idx, off) in enumerate(zip(self.groupers, self.mlps, packs, idxs, offs)):
            if idx is None:
                with _stage("ball_query"):
                    idx = pointnet2_utils.ball_query(grouper.radius, grouper.nsample, xyz, new_xyz)
            with _stage("sa_mlp"):
                # few columns (one frame per call, deep level): one launch per layer with one wave per 32 x 32 tile
                # fills the chip where the fused chain has a handful of 64-column workgroups
                small = None
                if xyz.size(0) * new_xyz.size(1) * grouper.nsample < 64 * _small_batch.MAX_FUSED_WGS:
                    small = _small_batch.folded_layers(mlp)
                if small is not None:
                    _small_batch.sa_scale(xyz, new_xyz, features, idx, grouper.use_xyz or features is None, small, out_pm,
                                          off)
                elif pre is not None:
                    if amax is not None:
                        amax._pvn3d_written = False
                    _ext.sa_mlp_maxpool(xyz, new_xyz, pre[si][0], idx, True, pre[si][1], out_pm, off, out_absmax=amax)
                    amax_writers += bool(amax is not None and amax._pvn3d_written)
                else:
                    if amax is not None:
                        amax._pvn3d_written = False
                    _ext.sa_mlp_maxpool(xyz, new_xyz, features, idx, grouper.use_xyz, packed, out_pm, off, out_absmax=amax)
                    amax_writers += bool(amax is not None and amax._pvn3d_written)
        out = out_pm[:, :, :sum(widths)].transpose(1, 2)
        if amax is not None and amax_writers == len(self.groupers) and getattr(self, "_point_major_out", False):
            _ext.seed_absmax(out, out_pm.size(0) * out_pm.size(1), sum(widths), out_pm.size(2), amax)
        # Stand-alone the module returns what the reference returns: a contiguous (B, C_out, npoint) tensor
        # (a caller may .view() it).  Pointnet2MSG marks its own levels `_point_major_out`: the next fused
        # level gathers rows from the point-major buffer in place, so the transposed view is handed over.
        return out if getattr(self, "_point_major_out", False) else out.contiguous()


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Set abstraction with multi-scale grouping.

    npoint: number of sampled centres; radii / nsamples / mlps: one entry per scale;
    ``mlps[i][0]`` is increased by 3 in place when use_xyz (reference :108-109).
    """

    def __init__(self, npoint, radii, nsamples, mlps, bn=True, use_xyz=True):
        super(PointnetSAModuleMSG, self).__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(
                pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                spec[0] += 3
            self.mlps.append(pt_utils.SharedMLP(spec, bn=bn))


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction (one radius / nsample / mlp)."""

    def __init__(self, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True):
        super(PointnetSAModule, self).__init__(
            mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz)


class PointnetFPModule(nn.Module):
    """Feature propagation: inverse-distance interpolation from `known` to `unknown`, then MLP."""

    def __init__(self, mlp, bn=True):
        super(PointnetFPModule, self).__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    @staticmethod
    def neighbours(unknown, known):
        """three_nn + inverse-distance weights (reference :183-186); depends on xyz only."""
        with _stage("three_nn"):
            if unknown.is_cuda and not torch.is_grad_enabled():
                # inference: the same fp32 formula as one kernel on three_nn's squared distances
                dist2, idx = _ext.three_nn(unknown, known)
                return idx, _ext.three_nn_weights(dist2)
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
        return idx, weight

    def forward(self, unknown, known, unknow_feats, known_feats, neighbours=None):
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n), known_feats (B,C2,m) -> (B,mlp[-1],n)"""
        if known is not None:
            idx, weight = neighbours if neighbours is not None else self.neighbours(unknown, known)
            if (FUSED_INFERENCE and not self.training and known_feats.is_cuda
                    and _no_grad_needed(unknow_feats, known_feats)
                    and not any(p.requires_grad and torch.is_grad_enabled() for p in self.parameters())):
                packed = _fused_mlp.pack_shared_mlp(self.mlp)
                if packed is not None and unknown.size(0) * unknown.size(1) < 64 * _small_batch.MAX_FUSED_WGS:
                    small = _small_batch.folded_layers(self.mlp)
                    if small is not None:
                        with _stage("fp_mlp"):
                            return _small_batch.fp_module(known_feats, unknow_feats, idx, weight.contiguous(), small,
                                                          getattr(self, "_point_major_out", False))
                if packed is not None:
                    with _stage("fp_mlp"):
                        return _ext.fp_interp_mlp(known_feats, unknow_feats, idx, weight.contiguous(), packed,
                                                  point_major_out=getattr(self, "_point_major_out", False))
            if (_train_mlp.train_fused_enabled() and self.training and known_feats.is_cuda and torch.is_grad_enabled()
                    and known_feats.dtype == torch.float32
                    and (unknow_feats is None or unknow_feats.dtype == torch.float32)):
                out = _train_mlp.fp_train(self, unknow_feats, known_feats, idx, weight)
                if out is not None:
                    return out
            interpolated = pointnet2_utils.three_interpolate(known_feats.contiguous(), idx, weight)
        else:
            interpolated = known_feats.expand(*(list(known_feats.size()[0:2]) + [unknown.size(1)]))
        if unknow_feats is not None:
            new_features = torch.cat([interpolated, unknow_feats], dim=1)
        else:
            new_features = interpolated
        return self.mlp(new_features.unsqueeze(-1)).squeeze(-1)
