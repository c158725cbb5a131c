"""Batched box / point transforms on device tensors for the motion-centric tracker.

Mirror of the tensor helpers of datasets/points_utils.py: `rotz_batch_tensor` :377-387,
`get_offset_points_tensor` :390-417, `get_offset_box_tensor` :420-436,
`remove_transform_points_tensor` :439-452.  Boxes are (B,4) = (x, y, z, yaw).  The reference mutates
its `points` argument in place (`points -= ...`); these functions do not (same values returned).
"""
import torch


def rotz_batch_tensor(t):
    """rotation matrices about z for angles t (...,) -> (...,3,3)"""
    c, s = torch.cos(t), torch.sin(t)
    zero, one = torch.zeros_like(c), torch.ones_like(c)
    rows = [torch.stack([c, -s, zero], -1), torch.stack([s, c, zero], -1), torch.stack([zero, zero, one], -1)]
    return torch.stack(rows, -2).to(torch.float32)


def get_offset_box_tensor(ref_box, offset_box):
    """box `ref_box` moved by `offset_box` expressed in the ref box frame: (B,4),(B,4) -> (B,4)"""
    rot = rotz_batch_tensor(ref_box[:, 3])
    center = torch.matmul(rot, offset_box[:, :3, None]).squeeze(-1) + ref_box[:, :3]
    return torch.cat([center, (ref_box[:, 3] + offset_box[:, 3])[:, None]], dim=-1)


def remove_transform_points_tensor(points, ref_box):
    """world -> frame of `ref_box`: points (B,N,3), ref_box (B,4)"""
    rot = rotz_batch_tensor(-ref_box[:, 3])
    return torch.matmul(points - ref_box[:, None, :3], rot.transpose(1, 2))


def get_offset_points_tensor(points, ref_box, offset_box):
    """apply the rigid motion `offset_box` (given in the frame of `ref_box`) to world points (B,N,3)"""
    rot = rotz_batch_tensor(-ref_box[:, 3])
    p = torch.matmul(points - ref_box[:, None, :3], rot.transpose(1, 2))          # into the box frame
    p = torch.matmul(p, rotz_batch_tensor(offset_box[:, 3]).transpose(1, 2)) + offset_box[:, None, :3]
    return torch.matmul(p, rot) + ref_box[:, None, :3]                             # back to the world
