"""`pointnet2_ops._ext` -- the nine operators the reference binds (see open3dsot_amd/ext.py)."""
from open3dsot_amd.ext import (  # noqa: F401
    furthest_point_sampling,
    gather_points,
    gather_points_grad,
    three_nn,
    three_interpolate,
    three_interpolate_grad,
    ball_query,
    group_points,
    group_points_grad,
)
