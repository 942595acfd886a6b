"""Operator API of the reference's pvn3d/lib/pointnet2_utils/pointnet2_utils.py, backed by the
gfx950 kernels in libpvn3d_hip.so.

Public names, argument order and return shapes are the reference's:
  furthest_point_sample(xyz, npoint)            (reference :37-64)
  gather_operation(features, idx)               (:67-101)
  three_nn(unknown, known) -> (dist, idx)       (:104-133)   dist = sqrt(dist2) as at :126
  three_interpolate(features, idx, weight)      (:136-190)
  grouping_operation(features, idx)             (:193-241)
  ball_query(radius, nsample, xyz, new_xyz)     (:244-273)   note the _ext argument order differs
  QueryAndGroup(radius, nsample, use_xyz)       (:276-330)
  GroupAll(use_xyz)                             (:333-376)
Differences, all additive: QueryAndGroup writes the concatenated (B,3+C,npoint,nsample) tensor
in one fused pass (no separate subtract / torch.cat) and accepts a precomputed ``idx``.
"""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from . import _ext

# Mixed-precision training (BASELINE config 5, bf16): inside a torch.autocast region the 1x1-conv GEMMs run
# in bf16 and hand bf16 activations to these ops.  The gather / scatter kernels are fp32 (like the
# reference's), so every differentiable op casts its floating inputs to fp32 on entry and runs with
# autocast off (custom_fwd / custom_bwd); indices and xyz are fp32 / int32 already.
_fwd32 = custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = custom_bwd(device_type="cuda")


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) -> (B,npoint) int32 indices of the iteratively farthest points."""
        out = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint) -> (B,C,npoint)"""
        ctx.save_for_backward(idx)
        ctx.n_src = features.size(2)
        return _ext.gather_points(features, idx)

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.gather_points_grad(grad_out.contiguous(), idx, ctx.n_src), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (B,n,3), known (B,m,3) -> dist (B,n,3) L2 distances, idx (B,n,3)"""
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, features, idx, weight):
        """features (B,c,m), idx/weight (B,n,3) -> (B,c,n)"""
        ctx.save_for_backward(idx, weight)
        ctx.m_src = features.size(2)
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        g = _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m_src)
        return g, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)"""
        ctx.save_for_backward(idx)
        ctx.n_src = features.size(2)
        return _ext.group_points(features, idx)

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.group_points_grad(grad_out.contiguous(), idx, ctx.n_src), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """xyz (B,N,3), new_xyz (B,npoint,3) -> (B,npoint,nsample) int32"""
        out = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class _GroupXyzFeatures(Function):
    """Fused gather + (xyz - centre) + concat; backward scatters to `features` (and xyz^T)."""

    @staticmethod
    @_fwd32
    def forward(ctx, xyz, new_xyz, features, idx, use_xyz):
        ctx.save_for_backward(idx)
        ctx.use_xyz = use_xyz
        ctx.n_src = xyz.size(1)
        ctx.has_feat = features is not None
        return _ext.group_xyz_features(xyz, new_xyz, features, idx, use_xyz)

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        c0 = 3 if ctx.use_xyz else 0
        g_xyz = g_new = g_feat = None
        if ctx.use_xyz and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            gx = grad_out[:, :3].contiguous()
            if ctx.needs_input_grad[0]:
                g_xyz = _ext.group_points_grad(gx, idx, ctx.n_src).transpose(1, 2).contiguous()
            if ctx.needs_input_grad[1]:
                g_new = -gx.sum(dim=3).transpose(1, 2).contiguous()
        if ctx.has_feat and ctx.needs_input_grad[2]:
            g_feat = _ext.group_points_grad(grad_out[:, c0:].contiguous(), idx, ctx.n_src)
        return g_xyz, g_new, g_feat, None, None


class QueryAndGroup(nn.Module):
    """Ball query of `radius` around each centre, then group xyz (centre-relative) and features.

    forward(xyz (B,N,3), new_xyz (B,npoint,3), features (B,C,N) or None)
        -> (B, 3 + C, npoint, nsample)   [C only when use_xyz is False]
    """

    def __init__(self, radius, nsample, use_xyz=True):
        super(QueryAndGroup, self).__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None, idx=None):
        if idx is None:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        return _GroupXyzFeatures.apply(xyz, new_xyz, features, idx, self.use_xyz)


class GroupAll(nn.Module):
    """Groups every point into a single neighbourhood: (B, 3 + C, 1, N)."""

    def __init__(self, use_xyz=True):
        super(GroupAll, self).__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped_features = features.unsqueeze(2)
        if self.use_xyz:
            return torch.cat([grouped_xyz, grouped_features], dim=1)
        return grouped_features
