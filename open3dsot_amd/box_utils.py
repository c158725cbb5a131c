"""Batched box / point transforms on device tensors for the motion-centric tracker.

Mirror of the tensor helpers of datasets/points_utils.py: `rotz_batch_tensor` :377-387,
`get_offset_points_tensor` :390-417, `get_offset_box_tensor` :420-436,
`remove_transform_points_tensor` :439-452.  Boxes are (B,4) = (x, y, z, yaw).  The reference mutates
its `points` argument in place (`points -= ...`); these functions do not (same values returned).
"""
import ctypes

import torch

from . import capi

_vp, _i, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
capi.register("o3d_motion_merge_fwd", [_vp, _l, _l, _vp, _vp, _i, _i, _vp, _vp, _vp])
capi.register("o3d_offset_box", [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp])
capi.register("o3d_motion_merge_bwd", [_vp, _l, _l, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp])

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
    # float32 like the reference's; fp64 angles stay fp64 (the tests evaluate these helpers in fp64 as the specification)
    return torch.stack(rows, -2).to(torch.float64 if t.dtype == torch.float64 else torch.float32)


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


class OffsetBox(torch.autograd.Function):
    """get_offset_box_tensor as one launch each way (csrc/boxcloud.hip::offset_box_kernel)"""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, ref, off):
        lib = capi.load()
        r, o = ref.detach().contiguous(), off.detach().contiguous()
        box = torch.empty_like(r)
        capi.check(lib.o3d_offset_box(r.data_ptr(), o.data_ptr(), r.shape[0], box.data_ptr(), None, None, None,
                                      torch.cuda.current_stream(r.device).cuda_stream), "offset_box")
        ctx.saved = (r, o)
        return box

    @staticmethod
    @capi.on_tensor_device
    def backward(ctx, g):
        lib = capi.load()
        r, o = ctx.saved
        g = g.contiguous()
        g_ref = torch.empty_like(r) if ctx.needs_input_grad[0] else None
        g_off = torch.empty_like(o) if ctx.needs_input_grad[1] else None
        capi.check(lib.o3d_offset_box(r.data_ptr(), o.data_ptr(), r.shape[0], None, g.data_ptr(),
                                      g_ref.data_ptr() if g_ref is not None else None, g_off.data_ptr() if g_off is not None else None,
                                      torch.cuda.current_stream(r.device).cuda_stream), "offset_box_bwd")
        return g_ref, g_off


def get_offset_box_tensor(ref_box, offset_box):
    """box `ref_box` moved by `offset_box` expressed in the ref box frame: (B,4),(B,4) -> (B,4)"""
    if (ref_box.is_cuda and ref_box.dtype == torch.float32 and offset_box.dtype == torch.float32 and ref_box.dim() == 2 and
            ref_box.shape == offset_box.shape and ref_box.shape[1] == 4 and ref_box.shape[0] > 0):
        return OffsetBox.apply(ref_box, offset_box)
    return get_offset_box_tensor_reference(ref_box, offset_box)


def get_offset_box_tensor_reference(ref_box, offset_box):
    """the torch-op form (the specification of OffsetBox; CPU and fp64 inputs)"""
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


class MotionMerge(torch.autograd.Function):
    """M2-Track between its stages (models/m2track.py:120-137) as one launch each way (csrc/boxcloud.hip):
    apply(points (B,C>=3,N) masked, channel-major, the first N/2 of the previous frame; prev (B,4) | None; motion (B,4))
      -> merged (B,3,N) = both halves in the frame of the first-stage box, aux (B,4) = that box
    i.e. aux = get_offset_box_tensor(prev, motion); merged = remove_transform_points_tensor(cat(get_offset_points_tensor(
    first half, prev, motion), second half), aux) in the (B,3,N) layout.  No gradient to the points (they are data times a
    hard mask in the model)."""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, points, prev, motion):
        lib = capi.load()
        B, C, N = points.shape
        pts = points.detach()
        if pts.stride(2) != 1:
            pts = pts.contiguous()
        pv = prev.detach().contiguous() if prev is not None else None
        mo = motion.detach().contiguous()
        merged = torch.empty((B, 3, N), device=pts.device, dtype=torch.float32)
        aux = torch.empty((B, 4), device=pts.device, dtype=torch.float32)
        st = torch.cuda.current_stream(pts.device).cuda_stream
        capi.check(lib.o3d_motion_merge_fwd(pts.data_ptr(), pts.stride(0), pts.stride(1), pv.data_ptr() if pv is not None else None,
                                            mo.data_ptr(), B, N, merged.data_ptr(), aux.data_ptr(), st), "motion_merge_fwd")
        ctx.saved = (pts, pv, mo)
        ctx.prev_grad = prev is not None and ctx.needs_input_grad[1]
        ctx.set_materialize_grads(False)
        return merged, aux

    @staticmethod
    @capi.on_tensor_device
    def backward(ctx, g_merged, g_aux):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("MotionMerge: no gradient with respect to the points")
        lib = capi.load()
        pts, pv, mo = ctx.saved
        B, _, N = pts.shape
        if g_merged is None:
            g_merged = torch.zeros((B, 3, N), device=pts.device, dtype=torch.float32)
        g_merged = g_merged.contiguous()
        g_aux = g_aux.contiguous() if g_aux is not None else None
        g_motion = torch.empty_like(mo)
        g_prev = torch.empty_like(pv) if ctx.prev_grad else None
        st = torch.cuda.current_stream(pts.device).cuda_stream
        capi.check(lib.o3d_motion_merge_bwd(pts.data_ptr(), pts.stride(0), pts.stride(1), pv.data_ptr() if pv is not None else None,
                                            mo.data_ptr(), B, N, g_merged.data_ptr(), g_aux.data_ptr() if g_aux is not None else None,
                                            g_prev.data_ptr() if g_prev is not None else None, g_motion.data_ptr(), st),
                   "motion_merge_bwd")
        return None, g_prev, g_motion


def motion_merge_supported(points, motion):
    return (points.is_cuda and points.dtype == torch.float32 and motion.dtype == torch.float32 and points.dim() == 3 and
            points.shape[1] >= 3 and points.shape[2] % 2 == 0 and not points.requires_grad)


def motion_merge_reference(points, prev, motion):
    """the same function from the reference's helpers (the specification of MotionMerge): -> merged (B,3,N), aux (B,4)"""
    N = points.shape[2]
    if prev is None:
        prev = torch.zeros_like(motion)
    xyz0, xyz1 = points[:, :3, :N // 2], points[:, :3, N // 2:]
    aux = get_offset_box_tensor_reference(prev, motion)
    moved = get_offset_points_tensor(xyz0.transpose(1, 2), prev, motion).transpose(1, 2)
    merged = torch.cat([moved, xyz1], dim=-1)
    return remove_transform_points_tensor(merged.transpose(1, 2), aux).transpose(1, 2), aux
