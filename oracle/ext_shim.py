"""CPU look-alike of `pointnet2_ops._ext` on torch tensors (TEST INFRASTRUCTURE).

Lets the reference's own Python layers (pointnet2/utils/*.py, models/head/*.py,
models/backbone/pointnet.py) be imported and run in the build container, where neither
the upstream CUDA extension nor a GPU exists: `install()` registers this module as
`pointnet2_ops._ext` in sys.modules.  Every function forwards to oracle/ops.py (the plain-C
restatement).  Used only by tests/golden/make_golden.py and the tests.
"""
import sys
import types

import numpy as np
import torch

from . import ops


def _np(t):
    return t.detach().cpu().contiguous().numpy()


def furthest_point_sampling(xyz, npoint):
    return torch.from_numpy(ops.furthest_point_sampling(_np(xyz), int(npoint)))


def gather_points(features, idx):
    return torch.from_numpy(ops.gather_points(_np(features), _np(idx)))


def gather_rows(src, idx):
    """src (B,N,D), idx (B,npoint) -> (B,npoint,D): the channel-major gather of pointnet2_utils.py:68-103 on the
    transposed tensor (pointnet2_modules.py:52-62 transposes around it)"""
    a = _np(src)
    out = ops.gather_points(a.transpose(0, 2, 1).copy(), _np(idx))
    return torch.from_numpy(out.transpose(0, 2, 1).copy())


def gather_points_grad(grad_out, idx, n):
    return torch.from_numpy(ops.gather_points_grad(_np(grad_out), _np(idx), int(n)))


def three_nn(unknown, known):
    d2, idx = ops.three_nn(_np(unknown), _np(known))
    return torch.from_numpy(d2), torch.from_numpy(idx)


def three_interpolate(features, idx, weight):
    return torch.from_numpy(ops.three_interpolate(_np(features), _np(idx), _np(weight)))


def three_interpolate_grad(grad_out, idx, weight, m):
    return torch.from_numpy(ops.three_interpolate_grad(_np(grad_out), _np(idx), _np(weight), int(m)))


def ball_query(new_xyz, xyz, radius, nsample):
    return torch.from_numpy(ops.ball_query(_np(new_xyz), _np(xyz), float(np.float32(radius)), int(nsample)))


def group_points(features, idx):
    return torch.from_numpy(ops.group_points(_np(features), _np(idx)))


def group_points_grad(grad_out, idx, n):
    return torch.from_numpy(ops.group_points_grad(_np(grad_out), _np(idx), int(n)))


def install():
    """Register this module as `pointnet2_ops._ext` (and a stub parent package)."""
    pkg = types.ModuleType("pointnet2_ops")
    pkg.__path__ = []
    me = sys.modules[__name__]
    pkg._ext = me
    sys.modules["pointnet2_ops"] = pkg
    sys.modules["pointnet2_ops._ext"] = me
    return me
