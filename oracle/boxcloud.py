"""CPU restatement (numpy, fp64) of the BoxCloud / input-regularisation helpers -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this; the product path
(open3dsot_amd/points_utils.py -> csrc/boxcloud.hip) never does.

Follows the reference line by line:
  box_corners                 datasets/data_classes.py:226-250   (Box.corners)
  get_point_to_box_distance   datasets/points_utils.py:127-143   (cdist of points vs [centre | corners])
  regularize_pc               datasets/points_utils.py:24-40     (numpy choice, with replacement when short,
                                                                  zeros when <= 2 points)
PINNED: tests/golden/ref_boxcloud.npz holds outputs of the reference's own functions run in this
container (tests/golden/make_golden_boxcloud.py); tests/test_boxcloud.py checks this file against them.
"""
import numpy as np


def box_corners(center, wlh, rot, wlh_factor=1.0):
    """-> (3, 8) corners, first four facing forward (data_classes.py:226-250)"""
    w, l, h = np.asarray(wlh, dtype=np.float64) * wlh_factor
    x = l / 2 * np.array([1, 1, 1, 1, -1, -1, -1, -1])
    y = w / 2 * np.array([1, -1, -1, 1, 1, -1, -1, 1])
    z = h / 2 * np.array([1, 1, -1, -1, 1, 1, -1, -1])
    corners = np.dot(np.asarray(rot, dtype=np.float64).reshape(3, 3), np.vstack((x, y, z)))
    return corners + np.asarray(center, dtype=np.float64).reshape(3, 1)


def get_point_to_box_distance(points, center, wlh, rot, wlh_factor=1.0):
    """points (N,3) -> (N,9) distances to [centre, corner 0..7] (points_utils.py:127-143)"""
    points = np.asarray(points, dtype=np.float64)
    assert points.shape[1] == 3
    landmarks = np.concatenate([np.asarray(center, dtype=np.float64).reshape(3, 1),
                                box_corners(center, wlh, rot, wlh_factor)], axis=1)      # (3, 9)
    diff = points[:, None, :] - landmarks.T[None, :, :]
    return np.sqrt((diff * diff).sum(-1))


def regularize_pc(points, sample_size, seed=None):
    """(points (n,3)) -> (resampled (sample_size,3), indices | None) (points_utils.py:24-40)"""
    num_points = points.shape[0]
    idx = None
    rng = np.random if seed is None else np.random.default_rng(seed)
    if num_points > 2:
        if num_points != sample_size:
            idx = rng.choice(num_points, size=sample_size, replace=sample_size > num_points)
        else:
            idx = np.arange(num_points)
    if idx is not None:
        return points[idx, :], idx
    return np.zeros((sample_size, 3), dtype="float32"), None
