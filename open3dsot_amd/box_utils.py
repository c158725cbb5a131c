"""Batched box / point transforms on device tensors for the motion-centric tracker.

Mirror of the tensor helpers of datasets/points_utils.py: `rotz_batch_tensor` :377-387,
`get_offset_points_tensor` :390-417, `get_offset_box_tensor` :420-436,
`remove_transform_points_tensor` :439-452.  Boxes are (B,4) = (x, y, z, yaw).  The reference mutates
its `points` argument in place (`points -= ...`); these functions do not (same values returned).
"""
import torch


_ROTZ = {}        # device -> (basis (2, 9), constant part (9,)) of the z rotation as a linear map of (cos, sin)


def _rotz_constants(dev):
    key = str(dev)
    hit = _ROTZ.get(key)
    if hit is None:
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            return None               # host data cannot be uploaded by a capturing stream: the stacked form below
        basis = torch.tensor([[1.0, 0, 0, 0, 1, 0, 0, 0, 0], [0, -1.0, 0, 1, 0, 0, 0, 0, 0]], device=dev)
        const = torch.tensor([0.0, 0, 0, 0, 0, 0, 0, 0, 1], device=dev)      # 1-D: the GEMM's bias epilogue, no broadcast copy
        hit = _ROTZ[key] = (basis, const)
    return hit


def rotz_batch_tensor_stacked(t):
    """the reference's formulation (datasets/points_utils.py:377-387): nine launches and a four-stack backward"""
    c, s = torch.cos(t), torch.sin(t)
    zero, one = torch.zeros_like(c), torch.ones_like(c)
    rows = [torch.stack([c, -s, zero], -1), torch.stack([s, c, zero], -1), torch.stack([zero, zero, one], -1)]
    return torch.stack(rows, -2).to(torch.float32)


def rotz_batch_tensor(t):
    """rotation matrices about z for angles t (...,) -> (...,3,3).
    The same nine entries as the stacked form, bit for bit (c*1 + s*0 + 0 and c*0 + s*(-1) + 0 are exact), written as
    ONE small GEMM [cos t, sin t] . basis + const: four launches instead of nine, and a backward of one GEMM instead of
    four stack-backwards with their zero fills (the M2-Track step builds five of these per forward)."""
    consts = _rotz_constants(t.device) if t.dtype == torch.float32 else None
    if consts is None:
        return rotz_batch_tensor_stacked(t)
    cs = torch.stack([torch.cos(t), torch.sin(t)], -1).reshape(-1, 2)
    return torch.addmm(consts[1], cs, consts[0]).view(*t.shape, 3, 3)


def _parts(box):
    """(B,4) -> centre (B,3), yaw (B,): ONE split node (its backward is one concatenation) where two slices would each
    cost a zero fill + a copy + an accumulation in the backward"""
    center, yaw = box.split([3, 1], dim=1)
    return center, yaw.squeeze(1)


def get_offset_box_tensor(ref_box, offset_box):
    """box `ref_box` moved by `offset_box` expressed in the ref box frame: (B,4),(B,4) -> (B,4)"""
    ref_c, ref_t = _parts(ref_box)
    off_c, off_t = _parts(offset_box)
    rot = rotz_batch_tensor(ref_t)
    center = torch.matmul(rot, off_c.unsqueeze(-1)).squeeze(-1) + ref_c
    return torch.cat([center, (ref_t + off_t)[:, None]], dim=-1)


def remove_transform_points_tensor(points, ref_box):
    """world -> frame of `ref_box`: points (B,N,3), ref_box (B,4)"""
    ref_c, ref_t = _parts(ref_box)
    rot = rotz_batch_tensor(-ref_t)
    return torch.matmul(points - ref_c[:, None, :], rot.transpose(1, 2))


def get_offset_points_tensor(points, ref_box, offset_box):
    """apply the rigid motion `offset_box` (given in the frame of `ref_box`) to world points (B,N,3)"""
    ref_c, ref_t = _parts(ref_box)
    off_c, off_t = _parts(offset_box)
    rot = rotz_batch_tensor(-ref_t)
    p = torch.matmul(points - ref_c[:, None, :], rot.transpose(1, 2))             # into the box frame
    p = torch.matmul(p, rotz_batch_tensor(off_t).transpose(1, 2)) + off_c[:, None, :]
    return torch.matmul(p, rot) + ref_c[:, None, :]                                # back to the world
