"""Device-side mirror of the two input-preparation helpers that sit on the critical path of every
tracked frame (models/bat.py:41-55 `prepare_input`) -- SURVEY.md section 8f-2.

  get_point_to_box_distance   datasets/points_utils.py:127-143  -> csrc/boxcloud.hip (one launch, clouds
                              already resident; the reference runs scipy cdist in fp64 on the host and
                              copies the result to the GPU)
  regularize_pc               datasets/points_utils.py:24-40    -> the index draw stays numpy's (same
                              generator, same call => the same indices as the reference); the row gather
                              runs on the device when the cloud lives there.
A box is passed as (center (3), wlh (3) = width/length/height, rot (3,3) rotation matrix) -- the three
attributes of datasets/data_classes.py::Box the reference reads (`center`, `wlh`,
`orientation.rotation_matrix`).
"""
import ctypes

import numpy as np
import torch

from . import capi

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
capi.register("o3d_boxcloud", [_vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp])


def _dev32(x, dev):
    return torch.as_tensor(np.asarray(x, dtype=np.float32) if not torch.is_tensor(x) else x, dtype=torch.float32,
                           device=dev).contiguous()


def get_point_to_box_distance(points, center, wlh, rot, wlh_factor=1.0):
    """points (N,3) or (B,N,3) on the GPU; center/wlh (3) or (B,3); rot (3,3) or (B,3,3) -> (N,9) / (B,N,9)"""
    if not points.is_cuda:
        raise RuntimeError("get_point_to_box_distance: CPU not supported (tensor must be a GPU tensor)")
    single = points.dim() == 2
    pts = (points.unsqueeze(0) if single else points).contiguous().float()
    B, N, three = pts.shape
    if three != 3:
        raise ValueError("points must be (..., 3)")
    dev = pts.device
    c = _dev32(center, dev).reshape(-1, 3)
    s = _dev32(wlh, dev).reshape(-1, 3)
    r = _dev32(rot, dev).reshape(-1, 9)
    if not (c.shape[0] == s.shape[0] == r.shape[0] == B):
        raise ValueError("one box per cloud expected")
    out = torch.empty((B, N, 9), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        rc = capi.load().o3d_boxcloud(pts.data_ptr(), c.data_ptr(), s.data_ptr(), r.data_ptr(), float(wlh_factor), B, N,
                                      out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        raise RuntimeError("o3d_boxcloud failed: %d" % rc)
    return out[0] if single else out


def regularize_pc(points, sample_size, seed=None):
    """points (n,3) tensor (any device) -> (resampled (sample_size,3) on the same device, indices | None).
    The indices are drawn exactly as the reference draws them (numpy, points_utils.py:24-40)."""
    num_points = points.shape[0]
    idx = None
    rng = np.random if seed is None else np.random.default_rng(seed)
    if num_points > 2:
        if num_points != sample_size:
            idx = rng.choice(num_points, size=sample_size, replace=sample_size > num_points)
        else:
            idx = np.arange(num_points)
    if idx is None:
        return torch.zeros((sample_size, 3), dtype=torch.float32, device=points.device), None
    return points.index_select(0, torch.as_tensor(idx, dtype=torch.long, device=points.device)), idx
