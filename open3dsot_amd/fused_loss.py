"""The trackers' loss (models/base_model.py:122-164, models/bat.py:57-65,131-137, models/p2b.py:69-74)
as ONE launch of csrc/loss.hip: the five loss terms, the weighted total and the gradients of the
total w.r.t. the network outputs.  `MatchingBaseModel.compute_loss` (torch ops, ~110 launches
forward+backward) stays as the specification the kernel is tested against (tests/test_model_gpu.py).
"""
import ctypes

import torch

from . import capi

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
capi.register("o3d_track_loss", [_vp] * 8 + [_i] * 4 + [_f] * 5 + [_vp] * 7)
capi.register("o3d_m2track_loss", [_vp] * 14 + [_i] * 3 + [_f] * 7 + [_vp] * 10)

_ON = {"on": True}
# (the total is its own 0-d output instead of `losses[0]`, whose backward would be a zero fill + a select copy; and there is no
# scaling launch when the upstream gradient is the constant 1 that DataParallelStep passes to `backward`)
_ONE = {}
KEYS = ("loss_objective", "loss_box", "loss_seg", "loss_vote", "loss_bc")


def set_fused_loss(enabled):
    _ON["on"] = bool(enabled)


def enabled():
    return _ON["on"]


def _ptr(t):
    return t.data_ptr() if t is not None else None


def one(dev):
    """the constant 1.0 (0-d, on `dev`) to hand to `loss.backward(gradient=...)`: autograd then launches no `ones_like`,
    and FusedTrackLoss.backward recognises it (by address) and skips its multiply.  Created on first use and kept:
    call it once outside a HIP-graph capture (DataParallelStep.__init__ does)."""
    key = str(torch.device(dev))
    t = _ONE.get(key)
    if t is None:
        t = _ONE[key] = torch.ones((), device=dev, dtype=torch.float32)
    return t


class FusedTrackLoss(torch.autograd.Function):
    """(weights, cla, vote, boxes, bc_pred | None, seg, box_label, centers, bc_label | None) -> (total (), losses[6])
    with losses = (total, objective, box, seg, vote, bc); only `total` carries gradient."""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, weights, cla, vote, boxes, bc_pred, seg, box_label, centers, bc_label):
        lib = capi.load()
        f32 = torch.float32
        tensors = [cla, vote, boxes, bc_pred, seg, box_label, centers, bc_label]
        cla, vote, boxes, bc_pred, seg, box_label, centers, bc_label = [
            t.detach().contiguous().to(f32) if t is not None else None for t in tensors]
        B, N = cla.shape
        P = boxes.shape[1]
        K = bc_pred.shape[2] if bc_pred is not None else 0
        dev = cla.device
        buf = torch.empty((512 + 6,), device=dev, dtype=f32)      # block sums + the six losses
        losses = buf[512:]
        need = any(ctx.needs_input_grad)
        grads = [torch.empty_like(cla), torch.empty_like(vote), torch.empty_like(boxes),
                 torch.empty_like(bc_pred) if bc_pred is not None else None] if need else [None] * 4
        st = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.o3d_track_loss(cla.data_ptr(), seg.data_ptr(), vote.data_ptr(), box_label.data_ptr(),
                                centers.data_ptr(), boxes.data_ptr(), _ptr(bc_pred), _ptr(bc_label), B, N, P, K,
                                *[float(w) for w in weights], buf.data_ptr(), losses.data_ptr(), *[_ptr(g) for g in grads], st)
        if rc != 0:
            raise RuntimeError("o3d_track_loss failed: %d" % rc)
        ctx.grads = grads
        ctx.mark_non_differentiable(losses)
        ctx.set_materialize_grads(False)
        return losses[0], losses

    @staticmethod
    @capi.on_tensor_device
    def backward(ctx, g, _g_parts=None):
        # kept: a second backward (retain_graph=True) scales the same tensors again.  NB the constant-1 path below hands out
        # THESE tensors (no copy): a consumer that modified its incoming gradient in place would change what a second
        # backward returns -- the fused stacks do not (they pack / read), and autograd's own accumulation is out of place here
        grads = ctx.grads
        if g is None:
            return (None,) * 9
        known = _ONE.get(str(g.device))
        if known is not None and g.data_ptr() == known.data_ptr() and g.dim() == 0:
            return (None, grads[0], grads[1], grads[2], grads[3], None, None, None, None)       # times the constant 1
        live = [t for t in grads if t is not None]
        scaled = iter(torch._foreach_mul(live, g))               # one launch; only the total carries gradient
        out = [next(scaled) if t is not None else None for t in grads]
        return (None, out[0], out[1], out[2], out[3], None, None, None, None)


def track_loss(config, data, output, with_bc):
    """-> (total, {loss_*: 0-d tensors}) from the same dicts MatchingBaseModel.compute_loss takes"""
    c = config
    weights = (c.objectiveness_weight, c.box_weight, c.seg_weight, c.vote_weight, c.bc_weight if with_bc else 0.0)
    total, losses = FusedTrackLoss.apply(weights, output["estimation_cla"], output["vote_xyz"], output["estimation_boxes"],
                                  output["pred_search_bc"] if with_bc else None, data["seg_label"], data["box_label"],
                                  output["center_xyz"], data["points2cc_dist_s"] if with_bc else None)
    parts = losses.detach()
    ld = {k: parts[i + 1] for i, k in enumerate(KEYS) if with_bc or k != "loss_bc"}
    return total, ld


M2_KEYS = ("loss_motion_cls", "loss_center", "loss_angle", "loss_center_prev", "loss_angle_prev", "loss_seg", "loss_center_aux",
           "loss_center_motion", "loss_angle_aux", "loss_angle_motion", "loss_bc")


class FusedM2Loss(torch.autograd.Function):
    """M2-Track's loss (models/m2track.py:153-231) as two launches of csrc/loss.hip:
    apply(weights (w_center, w_angle, w_seg, w_bc, w_mcls), seg_logits (B,2,N), pred_bc (B,N,K) | None, motion_cls (B,2) | None,
    motion_pred (B,4), aux (B,4), est (B,4) | None, prev (B,4) | None, seg_label, prev_bc, this_bc, state | None, motion_label,
    box_label, box_label_prev) -> (total (), losses[12]); only `total` carries gradient.  The torch-op `compute_loss_reference`
    of open3dsot_amd/m2track.py stays as the specification (tests/test_model_gpu.py::test_fused_m2_loss_...)."""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, weights, seg_logits, pred_bc, motion_cls, motion_pred, aux, est, prev, seg_label, prev_bc, this_bc, state,
                motion_label, box_label, box_label_prev):
        lib = capi.load()
        f32 = torch.float32
        dev = seg_logits.device

        def fl(t):
            return t.detach().contiguous().to(f32) if t is not None else None

        def lg(t):
            return t.detach().contiguous().to(torch.int64) if t is not None else None
        seg_logits, pred_bc, motion_cls, motion_pred, aux, est, prev = [fl(t) for t in (seg_logits, pred_bc, motion_cls,
                                                                                        motion_pred, aux, est, prev)]
        prev_bc, this_bc, motion_label, box_label, box_label_prev = [fl(t) for t in (prev_bc, this_bc, motion_label, box_label,
                                                                                     box_label_prev)]
        seg_label, state = lg(seg_label), lg(state)
        B, _, N = seg_logits.shape
        K = pred_bc.shape[2] if pred_bc is not None else 0
        buf = torch.empty((64 * 16 + 12,), device=dev, dtype=f32)
        losses = buf[64 * 16:]
        need = any(ctx.needs_input_grad)
        preds = (seg_logits, pred_bc, motion_cls, motion_pred, aux, est, prev)
        grads = [torch.empty_like(t) if (need and t is not None) else None for t in preds]
        st = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.o3d_m2track_loss(seg_logits.data_ptr(), seg_label.data_ptr(), _ptr(pred_bc), _ptr(prev_bc), _ptr(this_bc),
                                  _ptr(motion_cls), _ptr(state), motion_pred.data_ptr(), motion_label.data_ptr(), aux.data_ptr(),
                                  _ptr(est), _ptr(prev), box_label.data_ptr(), _ptr(box_label_prev), B, N, K,
                                  *[float(w) for w in weights], 0.5, 2.0, buf.data_ptr(), losses.data_ptr(),
                                  *[_ptr(g) for g in grads], st)
        if rc != 0:
            raise RuntimeError("o3d_m2track_loss failed: %d" % rc)
        ctx.grads = grads
        ctx.mark_non_differentiable(losses)
        ctx.set_materialize_grads(False)
        return losses[0], losses

    @staticmethod
    @capi.on_tensor_device
    def backward(ctx, g, _g_parts=None):
        grads = ctx.grads          # (kept for a second backward; handed out uncopied under the constant-1 seed: see FusedTrackLoss)
        if g is None:
            return (None,) * 15
        known = _ONE.get(str(g.device))
        if not (known is not None and g.data_ptr() == known.data_ptr() and g.dim() == 0):
            live = [t for t in grads if t is not None]
            scaled = iter(torch._foreach_mul(live, g))
            grads = [next(scaled) if t is not None else None for t in grads]
        return (None, *grads, None, None, None, None, None, None, None)


def m2track_loss(config, data, output, use_motion_cls, use_second_stage, use_prev_refinement, box_aware):
    """-> {loss_*: 0-d tensors} with the keys of M2TRACK.compute_loss (only loss_total carries gradient)"""
    c = config
    if box_aware and not (data["prev_bc"].shape == data["this_bc"].shape and
                          2 * data["prev_bc"].shape[1] == output["pred_bc"].shape[1]):
        raise ValueError("m2track_loss: prev_bc / this_bc must be the two halves of pred_bc's points")
    weights = (c.center_weight, c.angle_weight, c.seg_weight, c.bc_weight if box_aware else 0.0,
               c.motion_cls_seg_weight if use_motion_cls else 0.0)
    total, losses = FusedM2Loss.apply(
        weights, output["seg_logits"], output["pred_bc"] if box_aware else None, output["motion_cls"] if use_motion_cls else None,
        output["motion_pred"], output["aux_estimation_boxes"], output["estimation_boxes"] if use_second_stage else None,
        output["estimation_boxes_prev"] if use_prev_refinement else None, data["seg_label"],
        data["prev_bc"] if box_aware else None, data["this_bc"] if box_aware else None,
        data["motion_state_label"] if use_motion_cls else None, data["motion_label"], data["box_label"], data["box_label_prev"])
    parts = losses.detach()
    ld = {"loss_total": total}
    for i, k in enumerate(M2_KEYS):
        ld[k] = parts[i + 1]
    if not use_motion_cls:
        del ld["loss_motion_cls"]
    if not use_second_stage:
        del ld["loss_center"], ld["loss_angle"]
    if not use_prev_refinement:
        del ld["loss_center_prev"], ld["loss_angle_prev"]
    if not box_aware:
        del ld["loss_bc"]
    return ld
