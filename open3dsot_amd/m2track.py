"""M2-Track: the motion-centric two-stage tracker (BASELINE config 4).

Host-side mirror (plain nn.Module, no Lightning) of models/m2track.py: `__init__` :18-71 (module
names = state_dict keys of the reference), `forward` :73-151, `compute_loss` :153-231 and the
training-step arithmetic :233-264.  It uses no pointnet2 operator: the per-point stacks
(`SegPointNet`, `MiniPointNet`, models/backbone/pointnet.py:91-204) are 1x1 Conv1d + BatchNorm1d +
ReLU + global max, run here as GEMMs on the flat (C, B*N) layout (open3dsot_amd/backbone.py).
Differences from the reference, by design: the class weights of the segmentation loss are created
on the logits' device (the reference calls `.cuda()`, :171); no torchmetrics / `.item()` logging.
"""
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

from . import box_utils
from .backbone import MiniPointNet, SegPointNet
from .nn_blocks import RowBatchNorm1d

# cfgs/M2_track_kitti.yaml (model / loss keys)
M2_KITTI = dict(net_model="m2track", box_aware=True, use_motion_cls=True, use_second_stage=True,
                use_prev_refinement=True, center_weight=2, angle_weight=10.0, seg_weight=0.1, bc_weight=1,
                motion_cls_seg_weight=0.1, point_sample_size=1024, optimizer="Adam", lr=0.001, wd=0,
                lr_decay_step=20, lr_decay_rate=0.1, batch_size=100)


_CONSTS = {}


def _const(dev, values):
    """a small constant vector on `dev`, uploaded once: host data inside the step would make it impossible to capture (the
    class weights of the segmentation loss were a `torch.tensor(..., device=...)` per call, and the HIP-graph capture of
    the M2-Track step failed on exactly that until round 3)"""
    key = (str(dev), values)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(values, device=dev, dtype=torch.float32)
    return t


def _head(out):
    return nn.Sequential(nn.Linear(256, 128), RowBatchNorm1d(128), nn.ReLU(),
                         nn.Linear(128, 128), RowBatchNorm1d(128), nn.ReLU(), nn.Linear(128, out))



class _PackParts(torch.autograd.Function):
    """torch.cat(parts, dim=1) of (B, C_i, N) tensors with any strides, written straight into the flat (sum C_i, B*N) layout the
    per-point stacks consume (csrc/heads.hip::pack_rows_kernel) and handed out as its (B, C, N) view: the stack takes that view
    as its operand without a copy (were a concatenation + a layout copy per stack input).  The backward is views of the
    incoming gradient."""

    @staticmethod
    def forward(ctx, *parts):
        from .fused_heads import pack_rows
        B, _, N = parts[0].shape
        ctx.sizes = [t.shape[1] for t in parts]
        rows = sum(ctx.sizes)
        with torch.cuda.device(parts[0].device):
            flat = pack_rows([t.detach() for t in parts], rows)
        return flat.view(rows, B, N).permute(1, 0, 2)

    @staticmethod
    def backward(ctx, g):
        return tuple(g.split(ctx.sizes, dim=1))


def _cat_channels(parts):
    """torch.cat(parts, dim=1); on the GPU one launch into the stacks' flat layout (see _PackParts)"""
    p0 = parts[0]
    # pack_rows takes B and N from the first part: a mismatched part must reach torch.cat, which raises
    if p0.is_cuda and len(parts) <= 4 and all(
            t.dtype == torch.float32 and t.dim() == 3 and t.device == p0.device and t.shape[0] == p0.shape[0]
            and t.shape[2] == p0.shape[2] for t in parts):
        return _PackParts.apply(*parts)
    return torch.cat(parts, dim=1)


class M2TRACK(nn.Module):
    def __init__(self, config=None, **kwargs):
        super().__init__()
        cfg = dict(M2_KITTI)
        cfg.update(kwargs)
        self.config = config if config is not None else SimpleNamespace(**cfg)
        c = self.config
        self.box_aware = getattr(c, "box_aware", False)
        self.use_motion_cls = getattr(c, "use_motion_cls", True)
        self.use_second_stage = getattr(c, "use_second_stage", True)
        self.use_prev_refinement = getattr(c, "use_prev_refinement", True)
        bc = 9 if self.box_aware else 0
        self.seg_pointnet = SegPointNet(input_channel=3 + 1 + 1 + bc, per_point_mlp1=[64, 64, 64, 128, 1024],
                                        per_point_mlp2=[512, 256, 128, 128], output_size=2 + bc)
        self.mini_pointnet = MiniPointNet(input_channel=3 + 1 + bc, per_point_mlp=[64, 128, 256, 512],
                                          hidden_mlp=[512, 256], output_size=-1)
        if self.use_second_stage:
            self.mini_pointnet2 = MiniPointNet(input_channel=3 + bc, per_point_mlp=[64, 128, 256, 512],
                                               hidden_mlp=[512, 256], output_size=-1)
            self.box_mlp = _head(4)
        if self.use_prev_refinement:
            self.final_mlp = _head(4)
        if self.use_motion_cls:
            self.motion_state_mlp = _head(2)
        self.motion_mlp = _head(4)

    def forward(self, input_dict):
        """points (B,N,5) [+ candidate_bc (B,N,9)] -> dict with estimation_boxes (B,4), seg_logits (B,2,N), ..."""
        from . import fused_heads
        # one forward = one scope: the prepared weight copies refreshed by one launch, every BatchNorm counter of the
        # forward (five stacks, twelve row BatchNorms) updated by one launch at the end
        with fused_heads.prep_scope(input_dict["points"].device, owner=self):
            return self._forward(input_dict)

    def _forward(self, input_dict):
        out = {}
        x = input_dict["points"].transpose(1, 2)
        if self.box_aware:
            x = _cat_channels([x, input_dict["candidate_bc"].transpose(1, 2)])
        seg_out = self.seg_pointnet(x)
        if self.box_aware:       # one split node instead of two slices (whose backward is 2 x (fill + copy) + an add)
            seg_logits, pred_bc = seg_out.split([2, seg_out.shape[1] - 2], dim=1)
        else:
            seg_logits = seg_out[:, :2, :]
        pred_cls = torch.argmax(seg_logits, dim=1, keepdim=True)                  # (B,1,N) hard mask, no gradient
        mask_points = x[:, :4, :] * pred_cls
        mask_xyz = mask_points                       # (B,4,N): data times the hard mask, no gradient; first N/2 = previous frame
        if self.box_aware:
            mask_pred_bc = pred_bc * pred_cls
            mask_points = torch.cat([mask_points, mask_pred_bc], dim=1)
            out["pred_bc"] = pred_bc.transpose(1, 2)
        from .fused_rows import seq_rows, seq_rows_group   # the heads' Linear -> BatchNorm1d -> ReLU rows on the GPU: one launch
        point_feature = self.mini_pointnet(mask_points)    # per layer, and the heads that read the same feature side by side
        heads = [self.motion_mlp] + ([self.motion_state_mlp] if self.use_motion_cls else []) + \
            ([self.final_mlp] if self.use_prev_refinement else [])
        head_out = seq_rows_group(heads, point_feature)
        motion_pred = head_out[0]                                                  # (B,4)
        if self.use_motion_cls:
            logits = head_out[1]
            motion_pred_masked = motion_pred * torch.argmax(logits, dim=1, keepdim=True)
            out["motion_cls"] = logits
        else:
            motion_pred_masked = motion_pred
        if self.use_prev_refinement:
            prev_boxes = head_out[-1]
            out["estimation_boxes_prev"] = prev_boxes[:, :4] if prev_boxes.shape[1] != 4 else prev_boxes
        else:
            prev_boxes = torch.zeros_like(motion_pred)
        if self.use_second_stage and box_utils.motion_merge_supported(mask_xyz, motion_pred_masked):
            # first-stage box + both halves of the points in its frame: one launch each way (csrc/boxcloud.hip)
            merged, aux_box = box_utils.MotionMerge.apply(mask_xyz, prev_boxes if self.use_prev_refinement else None,
                                                          motion_pred_masked)
        else:
            aux_box = box_utils.get_offset_box_tensor(prev_boxes, motion_pred_masked)  # first-stage box
            if self.use_second_stage:
                merged = box_utils.motion_merge_reference(mask_xyz, prev_boxes, motion_pred_masked)[0]
        if self.use_second_stage:
            if self.box_aware:
                merged = _cat_channels([merged, mask_pred_bc])
            offset = seq_rows(self.box_mlp, self.mini_pointnet2(merged))
            out["estimation_boxes"] = box_utils.get_offset_box_tensor(aux_box, offset)
        else:
            out["estimation_boxes"] = aux_box
        out.update({"seg_logits": seg_logits, "motion_pred": motion_pred, "aux_estimation_boxes": aux_box})
        return out

    def compute_loss(self, data, output):
        """models/m2track.py:153-231.  Same terms, same weights, same dict; the four (centre, angle) pairs -- first-stage
        box, refined box, previous-frame box, motion -- are evaluated as ONE stacked smooth-L1 each and the weighted total
        as one dot product, about half the launches of the term-by-term form (`compute_loss_reference`, which it equals
        to fp32 rounding: tests/test_golden_m2track.py)."""
        c = self.config
        aux, motion_pred, seg_logits = output["aux_estimation_boxes"], output["motion_pred"], output["seg_logits"]
        if seg_logits.is_cuda and seg_logits.dtype == torch.float32:
            from . import fused_loss
            if fused_loss.enabled():      # the whole loss and its gradients as two launches (csrc/loss.hip, round 4)
                return fused_loss.m2track_loss(c, data, output, self.use_motion_cls, self.use_second_stage,
                                               self.use_prev_refinement, self.box_aware)
        state = data["motion_state_label"]
        dev = seg_logits.device
        # rows of the stacked problem: (name suffix, prediction (B,4), label (B,4)); the motion row last
        rows = [("aux", aux, data["box_label"])]
        if self.use_second_stage:
            rows.append(("", output["estimation_boxes"], data["box_label"]))
        if self.use_prev_refinement:
            rows.append(("prev", output["estimation_boxes_prev"], data["box_label_prev"]))
        rows.append(("motion", motion_pred, data["motion_label"]))
        K, B = len(rows), aux.shape[0]
        pred = torch.stack([r[1] for r in rows])                                        # (K,B,4)
        label = torch.stack([r[2] for r in rows])
        pred_c, pred_t = pred.split([3, 1], dim=2)          # one split node: the backward is one concatenation
        per_center = F.smooth_l1_loss(pred_c, label[..., :3], reduction="none").mean(dim=2)                   # (K,B)
        per_angle = F.smooth_l1_loss(torch.sin(pred_t.squeeze(2)), torch.sin(label[..., 3]), reduction="none")  # (K,B)
        per = torch.stack([per_center, per_angle])                                      # (2,K,B)
        if self.use_motion_cls:      # the motion row averages over the moving samples only (:196-203), the others over B
            moving = state / (state.sum() + 1e-6)
            sample_w = torch.cat([torch.full((K - 1, B), 1.0 / B, device=dev, dtype=per.dtype), moving[None].to(per.dtype)])
            terms = (per * sample_w).sum(dim=2)                                         # (2,K)
        else:
            terms = per.mean(dim=2)
        ld, names, weights = {}, [], []
        for k, (suffix, _, _) in enumerate(rows):
            tag = "_" + suffix if suffix else ""
            ld["loss_center" + tag], ld["loss_angle" + tag] = terms[0, k], terms[1, k]
        extra = []
        loss_seg = F.cross_entropy(seg_logits, data["seg_label"], weight=_const(dev, (0.5, 2.0)).to(seg_logits.dtype))
        ld["loss_seg"] = loss_seg
        extra.append((loss_seg, c.seg_weight))
        if self.use_motion_cls:
            ld["loss_motion_cls"] = F.cross_entropy(output["motion_cls"], state)
            extra.append((ld["loss_motion_cls"], c.motion_cls_seg_weight))
        if self.box_aware:
            bc_label = torch.cat([data["prev_bc"], data["this_bc"]], dim=1)
            ld["loss_bc"] = F.smooth_l1_loss(output["pred_bc"], bc_label)
            extra.append((ld["loss_bc"], c.bc_weight))
        wvec = _const(dev, tuple([float(c.center_weight)] * K + [float(c.angle_weight)] * K + [float(w) for _, w in extra]))
        allterms = torch.cat([terms.reshape(-1), torch.stack([t for t, _ in extra])])
        ld["loss_total"] = torch.dot(allterms, wvec.to(allterms.dtype))
        return ld

    def compute_loss_reference(self, data, output):
        """the loss term by term as models/m2track.py:153-231 writes it (~60 launches forward, ~120 backward): the
        specification `compute_loss` is tested against"""
        c = self.config
        total = 0.0
        ld = {}
        aux, motion_pred, seg_logits = output["aux_estimation_boxes"], output["motion_pred"], output["seg_logits"]
        box_label, box_prev, motion_label = data["box_label"], data["box_label_prev"], data["motion_label"]
        state = data["motion_state_label"]
        center, angle = box_label[:, :3], torch.sin(box_label[:, 3])
        center_prev, angle_prev = box_prev[:, :3], torch.sin(box_prev[:, 3])
        center_motion, angle_motion = motion_label[:, :3], torch.sin(motion_label[:, 3])
        cls_w = _const(seg_logits.device, (0.5, 2.0)).to(seg_logits.dtype)
        loss_seg = F.cross_entropy(seg_logits, data["seg_label"], weight=cls_w)
        if self.use_motion_cls:
            loss_cls = F.cross_entropy(output["motion_cls"], state)
            total = total + loss_cls * c.motion_cls_seg_weight
            ld["loss_motion_cls"] = loss_cls
            lc = F.smooth_l1_loss(motion_pred[:, :3], center_motion, reduction="none")
            loss_center_motion = (state * lc.mean(dim=1)).sum() / (state.sum() + 1e-6)
            la = F.smooth_l1_loss(torch.sin(motion_pred[:, 3]), angle_motion, reduction="none")
            loss_angle_motion = (state * la).sum() / (state.sum() + 1e-6)
        else:
            loss_center_motion = F.smooth_l1_loss(motion_pred[:, :3], center_motion)
            loss_angle_motion = F.smooth_l1_loss(torch.sin(motion_pred[:, 3]), angle_motion)
        if self.use_second_stage:
            est = output["estimation_boxes"]
            ld["loss_center"] = F.smooth_l1_loss(est[:, :3], center)
            ld["loss_angle"] = F.smooth_l1_loss(torch.sin(est[:, 3]), angle)
            total = total + ld["loss_center"] * c.center_weight + ld["loss_angle"] * c.angle_weight
        if self.use_prev_refinement:
            est = output["estimation_boxes_prev"]
            ld["loss_center_prev"] = F.smooth_l1_loss(est[:, :3], center_prev)
            ld["loss_angle_prev"] = F.smooth_l1_loss(torch.sin(est[:, 3]), angle_prev)
            total = total + ld["loss_center_prev"] * c.center_weight + ld["loss_angle_prev"] * c.angle_weight
        loss_center_aux = F.smooth_l1_loss(aux[:, :3], center)
        loss_angle_aux = F.smooth_l1_loss(torch.sin(aux[:, 3]), angle)
        total = (total + loss_seg * c.seg_weight + loss_center_aux * c.center_weight + loss_angle_aux * c.angle_weight
                 + loss_center_motion * c.center_weight + loss_angle_motion * c.angle_weight)
        ld.update({"loss_seg": loss_seg, "loss_center_aux": loss_center_aux, "loss_center_motion": loss_center_motion,
                   "loss_angle_aux": loss_angle_aux, "loss_angle_motion": loss_angle_motion})
        if self.box_aware:
            bc_label = torch.cat([data["prev_bc"], data["this_bc"]], dim=1)
            ld["loss_bc"] = F.smooth_l1_loss(output["pred_bc"], bc_label)
            total = total + ld["loss_bc"] * c.bc_weight
        ld["loss_total"] = total
        return ld

    def training_loss(self, batch):
        """forward + losses (m2track.py:233-264 minus logging); returns (loss, loss_dict)"""
        ld = self.compute_loss(batch, self(batch))
        return ld["loss_total"], ld

    def configure_optimizers(self):
        c = self.config
        # same update rule as the reference's torch.optim.Adam; on the GPU one launch on flat buffers
        from . import optim
        opt = optim.make_adam(self.parameters(), c.lr, c.wd)
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=c.lr_decay_step, gamma=c.lr_decay_rate)
        return {"optimizer": opt, "lr_scheduler": sched}
