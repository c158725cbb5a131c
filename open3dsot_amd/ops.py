"""Autograd operators and grouping modules -- the Python operator boundary of the hot path.

Mirrors the public names, argument order, return arity and dtypes of the reference's
pointnet2/utils/pointnet2_utils.py (RandomDropout :24, FurthestPointSampling :35, GatherOperation :68,
ThreeNN :105, ThreeInterpolate :137, GroupingOperation :194, BallQuery :245,
QueryAndGroup :280, GroupAll :342, knn_point :388) so models written against that module
run unchanged.  The arithmetic is in libo3dsot_hip.so (open3dsot_amd.ext); nothing here
computes on the CPU.
"""
import torch
from torch import nn
from torch.autograd import Function

from . import ext as _ext


class RandomDropout(nn.Module):
    """pointnet2_utils.py:24-32: feature dropout with a drop probability drawn from U(0, p) per call.  (The reference passes
    the bound method `self.train` as the training flag -- always true -- and calls a function its pytorch_utils.py lacks;
    here the flag is `self.training` and the function is nn_blocks.feature_dropout_no_scaling.)"""

    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self.p = p
        self.inplace = inplace

    def forward(self, X):
        from .nn_blocks import feature_dropout_no_scaling
        theta = torch.Tensor(1).uniform_(0, self.p)[0]
        return feature_dropout_no_scaling(X, theta, self.training, self.inplace)


class FurthestPointSampling(Function):
    """(B,N,3) f32, npoint -> (B,npoint) i32; not differentiable."""

    @staticmethod
    def forward(ctx, xyz, npoint):
        out = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


def furthest_point_sample_pair(xyz_a, npoint_a, xyz_b, npoint_b):
    """furthest_point_sample of two independent sets of clouds in one launch (indices: not differentiable)"""
    with torch.no_grad():
        return _ext.furthest_point_sampling_pair(xyz_a.contiguous(), npoint_a, xyz_b.contiguous(), npoint_b)


class GatherOperation(Function):
    """features (B,C,N), idx (B,npoint) i32 -> (B,C,npoint); grad scatters back into (B,C,N)."""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.n_src = features.size(2)
        ctx.save_for_backward(idx)
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.gather_points_grad(grad_out.contiguous(), idx, ctx.n_src), None


gather_operation = GatherOperation.apply


def gather_xyz(xyz, idx):
    """new_xyz (B,npoint,3) = gather_operation(xyz^T, idx)^T (pointnet2_modules.py:52-62).  One launch on the
    point-major tensor when no gradient flows to the coordinates (they are inputs in every tracker); the reference's
    composition otherwise."""
    if xyz.requires_grad and torch.is_grad_enabled():
        return gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    return _ext.gather_rows(xyz.contiguous(), idx)


class ThreeNN(Function):
    """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) L2 distances, idx (B,n,3) i32)."""

    @staticmethod
    def forward(ctx, unknown, known):
        dist2, idx = _ext.three_nn(unknown, known)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """features (B,c,m), idx (B,n,3), weight (B,n,3) -> (B,c,n)."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.m_src = features.size(2)
        ctx.save_for_backward(idx, weight)
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        g = _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m_src)
        return g, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """features (B,C,N), idx (B,npoint,nsample) i32 -> (B,C,npoint,nsample)."""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.n_src = features.size(2)
        ctx.save_for_backward(idx)
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.group_points_grad(grad_out.contiguous(), idx, ctx.n_src), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """(radius, nsample, xyz (B,N,3), new_xyz (B,npoint,3)) -> (B,npoint,nsample) i32.

    Note the wrapper takes (xyz, new_xyz) while the native entry takes (new_xyz, xyz) --
    same as the reference (pointnet2_utils.py:247 vs :268)."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        out = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """Ball-query grouping: (xyz (B,N,3), new_xyz (B,npoint,3), features (B,C,N) | None)
    -> (B, 3+C, npoint, nsample), relative xyz first  [pointnet2_utils.py:299-339]."""

    def __init__(self, radius, nsample, use_xyz=True, return_idx=False, normalize_xyz=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.return_idx = return_idx
        self.normalize_xyz = normalize_xyz

    def query(self, xyz, new_xyz):
        return ball_query(self.radius, self.nsample, xyz, new_xyz)

    def forward(self, xyz, new_xyz, features=None):
        idx = self.query(xyz, new_xyz)
        rel = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        rel = rel - new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            rel = rel / self.radius
        if features is None:
            if not self.use_xyz:
                raise AssertionError("Cannot have not features and not use xyz as a feature!")
            out = rel
        else:
            grouped = grouping_operation(features, idx)
            out = torch.cat([rel, grouped], dim=1) if self.use_xyz else grouped
        return (out, idx) if self.return_idx else out


class GroupAll(nn.Module):
    """Single group holding every point: -> (B, 3+C, 1, N)  [pointnet2_utils.py:342-385]."""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        g_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return g_xyz
        g_feat = features.unsqueeze(2)
        return torch.cat([g_xyz, g_feat], dim=1) if self.use_xyz else g_feat


def knn_point(k, points1, points2):
    """k nearest neighbours of every points1 row among points2: (B,n1,k) i32.

    Reference: cdist + argsort[..., :k] (pointnet2_utils.py:388-402), whose tie order is
    unspecified; the HIP kernel selects by squared distance, ties -> lowest index."""
    return _ext.knn(points1.contiguous(), points2.contiguous(), k)
