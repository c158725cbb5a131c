"""BAT and P2B siamese trackers: forward graph, losses and the training-step arithmetic.

Host-side mirror (plain nn.Module, no Lightning) of
  models/bat.py   BAT.__init__ :18-38, forward :67-112, compute_loss :57-65, training_step :114-143
  models/p2b.py   P2B.__init__ :13-26, forward :28-59, training_step :61-78
  models/base_model.py  MatchingBaseModel.compute_loss :122-164, configure_optimizers :28-36
Submodule names (`backbone`, `conv_final`, `mlp_bc`, `xcorr`, `rpn`) and therefore
state_dict keys equal the reference's.  What is deliberately absent: the six `.item()`
host synchronisations per step that the reference spends on logging (bat.py:146-164) --
`training_loss` returns device tensors only.
"""
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

import ctypes

from . import capi, fused, fused_heads, fused_loss, optim
from . import nn_blocks as pt_utils
from .backbone import Pointnet_Backbone
from .rpn import P2BVoteNetRPN
from .xcorr import BoxAwareXCorr, P2B_XCorr

# cfgs/BAT_Car.yaml :9-10,27-37,40-44,53-61 and cfgs/P2B_Car.yaml (model/loss/optimizer keys)
BAT_CAR = dict(net_model="BAT", template_size=512, search_size=1024, use_fps=True, normalize_xyz=False,
               feature_channel=256, hidden_channel=256, out_channel=256, vote_channel=256,
               num_proposal=64, k=4, use_search_bc=False, use_search_feature=False, bc_channel=9,
               objectiveness_weight=1.5, box_weight=0.2, vote_weight=1.0, seg_weight=0.2, bc_weight=1.0,
               optimizer="Adam", lr=0.001, wd=0, lr_decay_step=12, lr_decay_rate=0.2, batch_size=50)
P2B_CAR = dict(net_model="P2B", template_size=512, search_size=1024, use_fps=False, normalize_xyz=False,
               feature_channel=256, hidden_channel=256, out_channel=256, vote_channel=256,
               num_proposal=64, objectiveness_weight=1.5, box_weight=0.2, vote_weight=1.0,
               seg_weight=0.2, optimizer="Adam", lr=0.001, wd=0, lr_decay_step=12,
               lr_decay_rate=0.2, batch_size=50)


def make_config(base, **overrides):
    cfg = dict(base)
    cfg.update(overrides)
    return SimpleNamespace(**cfg)


capi.register("o3d_best_proposal", [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p])


def best_proposal(boxes):
    """boxes (B,P,5) -> (boxes[b, argmax_p boxes[b,p,4], 0:4] (B,4), argmax (B,) int32)   [models/base_model.py:47-52]"""
    if not boxes.is_cuda:            # the CPU mirror used by the tests: the reference's own numpy formulation
        idx = boxes[:, :, 4].detach().cpu().numpy().argmax(axis=1)
        idx_t = torch.from_numpy(idx.astype("int32"))
        return boxes[torch.arange(boxes.shape[0]), idx_t.long(), :4], idx_t
    b = boxes.detach().contiguous().float()
    B, P, _ = b.shape
    out = torch.empty((B, 4), device=b.device, dtype=torch.float32)
    idx = torch.empty((B,), device=b.device, dtype=torch.int32)
    with torch.cuda.device(b.device):
        capi.check(capi.load().o3d_best_proposal(b.data_ptr(), B, P, out.data_ptr(), idx.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream), "best_proposal")
    return out, idx


def _seed_rows(a, b, sample_idxs, n):
    """a[:, idx] (, b[:, idx]) with idx = sample_idxs[:, :n]: the labels / BoxCloud rows of the points the backbone kept
    (models/bat.py:96-97,132-133, models/p2b.py:62-63: `.long()`, expand and one torch.gather per tensor).  On the GPU one
    launch for both tensors, straight from the int32 sampling indices (ext.gather_rows2); labels carry no gradient."""
    if a.is_cuda and a.dtype == torch.float32 and sample_idxs.dtype == torch.int32 and not a.requires_grad \
            and (b is None or (b.dtype == torch.float32 and not b.requires_grad)):
        from . import ext
        a3 = a.contiguous().unsqueeze(-1) if a.dim() == 2 else a.contiguous()
        ra, rb = ext.gather_rows2(a3, b.contiguous() if b is not None else None, sample_idxs, n)
        return (ra.squeeze(-1) if a.dim() == 2 else ra), rb
    idx = sample_idxs[:, :n].long()
    ra = a.gather(1, idx if a.dim() == 2 else idx[:, :, None].expand(-1, -1, a.shape[2]))
    rb = b.gather(1, idx[:, :, None].expand(-1, -1, b.shape[2])) if b is not None else None
    return ra, rb


# the coordinate-only part of the backbone (centres, ball queries, distinct-neighbour layouts of all three levels) computed with
# the sampling indices ahead of the step (`sampling_inputs`); False = inside the step, the route it is tested against
_GEOMETRY_PREFETCH = {"on": True, "min_batch": 8, "without_fps": False}


def set_geometry_prefetch(enabled):
    _GEOMETRY_PREFETCH["on"] = bool(enabled)


class MatchingBaseModel(nn.Module):
    def __init__(self, config=None, **kwargs):
        super().__init__()
        self.config = config if config is not None else SimpleNamespace(**kwargs)

    def configure_optimizers(self):
        c = self.config
        if str(c.optimizer).lower() == "sgd":
            opt = torch.optim.SGD(self.parameters(), lr=c.lr, momentum=0.9, weight_decay=c.wd)
        else:
            # same update rule as the reference's torch.optim.Adam; on the GPU one launch on flat buffers
            # (open3dsot_amd/optim.py::FlatAdam) instead of ~12 foreach launches per step
            opt = optim.make_adam(self.parameters(), c.lr, c.wd)
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=c.lr_decay_step, gamma=c.lr_decay_rate)
        return {"optimizer": opt, "lr_scheduler": sched}

    def sampling_inputs(self, batch, out=None):
        """{"fps_idx_t", "fps_idx_s"}: the level-0 farthest-point-sampling indices of a batch, or {} when the backbone
        does not sample (P2B); `out`: {key: tensor} destinations the larger ones are written into in place.  They depend
        on the input clouds only, so a training loop may compute them for batch t+1
        while step t runs (open3dsot_amd/dist.py::DataParallelStep.step(next_batch=...)): forward takes them from the
        input dict instead of launching the 766 serial FPS rounds at the head of the step."""
        t, s = batch["template_points"], batch["search_points"]
        idx = self.backbone.sampling_indices(t, t.shape[1] // 2, s, s.shape[1] // 2)
        given, out = out, ({} if idx is None else {"fps_idx_t": idx[0], "fps_idx_s": idx[1]})     # (`out` from here on: the result)
        # round 6: the rest of the backbone that depends on the coordinates only -- every level's centres, ball queries and
        # distinct-neighbour layout ("geo<level>.<name>", open3dsot_amd/fused.py::pair_geometry): 12 launches off the step's chain
        # (not for tiny batches: at batch 1 the step is 1.4 ms and HOST-bound, the ~35 extra eager launches of the prefetch
        # cost more than the 12 launches they take out of the graph -- p2b_batch1 1.39 -> 1.55 ms, profiles/r06_ab_geometry_prefetch.txt)
        # (and only beside a farthest-point sampling that is prefetched anyway: P2B, whose backbone does not sample, has no
        # prefetch stream at all without it, and gaining 12 launches cost it +0.1 ms of a larger input copy and a second
        # stream: 8.76 against 8.65 ms per step, gpurun_out/secondary_ab.txt)
        if (_GEOMETRY_PREFETCH["on"] and self.training and t.is_cuda and t.shape[0] >= _GEOMETRY_PREFETCH["min_batch"]
                and (idx is not None or _GEOMETRY_PREFETCH["without_fps"])):
            M, N = t.shape[1], s.shape[1]
            dst = None
            if given is not None and "geo0.gp" in given:      # destination buffers (a FlatBatch's own fields): written in place
                from .fused import GEO_KEYS
                dst = [{k: given["geo%d.%s" % (i, k)] for k in GEO_KEYS if "geo%d.%s" % (i, k) in given} for i in range(3)]
            geo = self.backbone.pair_geometry(t, [M // 2, M // 4, M // 8], s, [N // 2, N // 4, N // 8], idx, dst=dst)
            if geo is not None:
                for i, g in enumerate(geo):
                    for k, v in g.items():
                        out["geo%d.%s" % (i, k)] = v
        return out

    @staticmethod
    def _given_sampling(input_dict):
        return (input_dict["fps_idx_t"], input_dict["fps_idx_s"]) if "fps_idx_t" in input_dict else None

    @staticmethod
    def _given_geometry(input_dict):
        if "geo0.gp" not in input_dict:
            return None
        from .fused import GEO_KEYS
        return [{k: input_dict["geo%d.%s" % (i, k)] for k in GEO_KEYS} for i in range(3)]

    def evaluate_one_sample(self, data_dict):
        """The network half of MatchingBaseModel.evaluate_one_sample (models/base_model.py:44-57) without its host round
        trip: forward, then the proposal with the highest objectness (column 4; numpy's argmax: the first maximum) and
        its (x, y, z, theta) picked ON THE DEVICE (csrc/heads.hip::best_proposal_kernel) -- the reference copies the
        (64, 5) proposals to the host for that.  -> (best (B,4) float32, index (B,) int32), device tensors; feeding
        `getOffsetBB` (datasets/points_utils.py, host geometry on Box objects) is the caller's business."""
        end_points = self(data_dict)
        return best_proposal(end_points["estimation_boxes"])

    def compute_loss(self, data, output):
        """Siamese matching losses  [models/base_model.py:122-164]."""
        boxes = output["estimation_boxes"]          # (B,P,5)
        cla = output["estimation_cla"]              # (B,N)
        seg_label = data["seg_label"]               # (B,N)
        box_label = data["box_label"]               # (B,4)
        centers = output["center_xyz"]              # (B,P,3)
        vote_xyz = output["vote_xyz"]               # (B,N,3)

        loss_seg = F.binary_cross_entropy_with_logits(cla, seg_label)

        loss_vote = F.smooth_l1_loss(vote_xyz, box_label[:, None, :3].expand_as(vote_xyz), reduction="none")
        loss_vote = (loss_vote.mean(2) * seg_label).sum() / (seg_label.sum() + 1e-06)

        dist = torch.sqrt(torch.sum((centers - box_label[:, None, :3]) ** 2, dim=-1) + 1e-6)  # (B,P)
        near = (dist < 0.3).float()
        objectness_label = near
        objectness_mask = torch.clamp(near + (dist > 0.6).float(), max=1.0)
        pos_weight = torch.full((1,), 2.0, device=dist.device, dtype=dist.dtype)  # no H2D copy: graph-capturable
        # NB the reference leaves the default reduction ('mean') here (base_model.py:149-152), so
        # the mask only rescales a scalar; mirrored as is.
        loss_objective = F.binary_cross_entropy_with_logits(boxes[:, :, 4], objectness_label,
                                                            pos_weight=pos_weight)
        loss_objective = torch.sum(loss_objective * objectness_mask) / (torch.sum(objectness_mask) + 1e-6)

        loss_box = F.smooth_l1_loss(boxes[:, :, :4], box_label[:, None, :4].expand_as(boxes[:, :, :4]),
                                    reduction="none")
        loss_box = torch.sum(loss_box.mean(2) * objectness_label) / (objectness_label.sum() + 1e-6)
        return {"loss_objective": loss_objective, "loss_box": loss_box, "loss_seg": loss_seg,
                "loss_vote": loss_vote}


class P2B(MatchingBaseModel):
    def __init__(self, config=None, **kwargs):
        super().__init__(config if config is not None else make_config(P2B_CAR, **kwargs))
        c = self.config
        self.backbone = Pointnet_Backbone(c.use_fps, c.normalize_xyz, return_intermediate=False)
        self.conv_final = nn.Conv1d(256, c.feature_channel, kernel_size=1)
        self.xcorr = P2B_XCorr(feature_channel=c.feature_channel, hidden_channel=c.hidden_channel,
                               out_channel=c.out_channel)
        self.rpn = P2BVoteNetRPN(c.feature_channel, vote_channel=c.vote_channel,
                                 num_proposal=c.num_proposal, normalize_xyz=c.normalize_xyz)

    def forward(self, input_dict):
        with fused_heads.prep_scope(input_dict["search_points"].device, owner=self):
            return self._forward(input_dict)

    def _forward(self, input_dict):
        template, search = input_dict["template_points"], input_dict["search_points"]
        M, N = template.shape[1], search.shape[1]
        (template_xyz, template_feature, _), (search_xyz, search_feature, sample_idxs) = self.backbone.forward_pair(
            template, [M // 2, M // 4, M // 8], search, [N // 2, N // 4, N // 8], self._given_sampling(input_dict),
            self._given_geometry(input_dict))
        template_feature, search_feature = pt_utils.pointwise_conv1d_pair(self.conv_final, template_feature, search_feature)
        fusion = self.xcorr(template_feature, search_feature, template_xyz)
        boxes, cla, vote_xyz, centers = self.rpn(search_xyz, fusion)
        return {"estimation_boxes": boxes, "vote_center": vote_xyz, "pred_seg_score": cla,
                "center_xyz": centers, "sample_idxs": sample_idxs, "estimation_cla": cla,
                "vote_xyz": vote_xyz}

    def training_loss(self, batch):
        """forward + label re-indexing + weighted loss (p2b.py:61-78); returns (loss, loss_dict)."""
        end_points = self(batch)
        n_seed = end_points["estimation_cla"].shape[1]
        data = dict(batch)
        data["seg_label"] = _seed_rows(batch["seg_label"], None, end_points["sample_idxs"], n_seed)[0]
        if end_points["estimation_cla"].is_cuda and fused_loss.enabled():
            return fused_loss.track_loss(self.config, data, end_points, with_bc=False)     # one launch (csrc/loss.hip)
        ld = self.compute_loss(data, end_points)
        c = self.config
        loss = (ld["loss_objective"] * c.objectiveness_weight + ld["loss_box"] * c.box_weight
                + ld["loss_seg"] * c.seg_weight + ld["loss_vote"] * c.vote_weight)
        return loss, ld


class BAT(MatchingBaseModel):
    def __init__(self, config=None, **kwargs):
        super().__init__(config if config is not None else make_config(BAT_CAR, **kwargs))
        c = self.config
        self.backbone = Pointnet_Backbone(c.use_fps, c.normalize_xyz, return_intermediate=False)
        self.conv_final = nn.Conv1d(256, c.feature_channel, kernel_size=1)
        self.mlp_bc = (pt_utils.Seq(3 + c.feature_channel)
                       .conv1d(c.feature_channel, bn=True)
                       .conv1d(c.feature_channel, bn=True)
                       .conv1d(c.bc_channel, activation=None))
        self.xcorr = BoxAwareXCorr(feature_channel=c.feature_channel, hidden_channel=c.hidden_channel,
                                   out_channel=c.out_channel, k=c.k, use_search_bc=c.use_search_bc,
                                   use_search_feature=c.use_search_feature, bc_channel=c.bc_channel)
        self.rpn = P2BVoteNetRPN(c.feature_channel, vote_channel=c.vote_channel,
                                 num_proposal=c.num_proposal, normalize_xyz=c.normalize_xyz)

    def prepare_input(self, template_points, search_points, template_box):
        """Inference input of one frame (models/bat.py:41-55) built on the device: both clouds resampled to the
        configured sizes (seed 1, the reference's draw) and the template BoxCloud from csrc/boxcloud.hip --
        no host round trip.  template_points / search_points: (n,3) GPU tensors; template_box = (center,
        wlh, rotation matrix) of the template's target box."""
        from . import points_utils
        tp, _ = points_utils.regularize_pc(template_points, self.config.template_size, seed=1)
        sp, _ = points_utils.regularize_pc(search_points, self.config.search_size, seed=1)
        tp, sp = tp.float(), sp.float()
        bc = points_utils.get_point_to_box_distance(tp, *template_box)
        return {"template_points": tp[None], "search_points": sp[None], "points2cc_dist_t": bc[None]}

    def compute_loss(self, data, output):
        out = super().compute_loss(data, output)
        loss_bc = F.smooth_l1_loss(output["pred_search_bc"], data["points2cc_dist_s"], reduction="none")
        seg_label = data["seg_label"]
        out["loss_bc"] = torch.sum(loss_bc.mean(2) * seg_label) / (seg_label.sum() + 1e-6)
        return out

    def forward(self, input_dict):
        with fused_heads.prep_scope(input_dict["search_points"].device, owner=self):
            return self._forward(input_dict)

    def _forward(self, input_dict):
        template, search = input_dict["template_points"], input_dict["search_points"]
        template_bc = input_dict["points2cc_dist_t"]
        M, N = template.shape[1], search.shape[1]
        (template_xyz, template_feature, sample_idxs_t), (search_xyz, search_feature, sample_idxs) = \
            self.backbone.forward_pair(template, [M // 2, M // 4, M // 8], search, [N // 2, N // 4, N // 8],
                                       self._given_sampling(input_dict), self._given_geometry(input_dict))
        template_feature, search_feature = pt_utils.pointwise_conv1d_pair(self.conv_final, template_feature, search_feature)
        pred_search_bc = pt_utils.seq_apply(self.mlp_bc, [search_xyz.transpose(1, 2), search_feature])
        pred_search_bc = pred_search_bc.transpose(1, 2)                                    # (B,N/8,9)
        if pred_search_bc.is_cuda:
            # its two consumers on the device (the kNN kernel and the fused loss) both want it point-major and dense: one
            # copy here instead of one in each
            pred_search_bc = pred_search_bc.contiguous()
        template_bc = _seed_rows(template_bc, None, sample_idxs_t, M // 8)[0]               # (B,M/8,9)
        fusion = self.xcorr(template_feature, search_feature, template_xyz, search_xyz, template_bc,
                            pred_search_bc)
        boxes, cla, vote_xyz, centers = self.rpn(search_xyz, fusion)
        return {"estimation_boxes": boxes, "vote_center": vote_xyz, "pred_seg_score": cla,
                "center_xyz": centers, "sample_idxs": sample_idxs, "estimation_cla": cla,
                "vote_xyz": vote_xyz, "pred_search_bc": pred_search_bc}

    def training_loss(self, batch):
        """forward + label re-indexing + weighted loss (bat.py:114-143); returns (loss, loss_dict)."""
        end_points = self(batch)
        n_seed = end_points["estimation_cla"].shape[1]
        data = dict(batch)
        data["seg_label"], data["points2cc_dist_s"] = _seed_rows(batch["seg_label"], batch["points2cc_dist_s"],
                                                                 end_points["sample_idxs"], n_seed)
        if end_points["estimation_cla"].is_cuda and fused_loss.enabled():
            return fused_loss.track_loss(self.config, data, end_points, with_bc=True)      # one launch (csrc/loss.hip)
        ld = self.compute_loss(data, end_points)
        c = self.config
        loss = (ld["loss_objective"] * c.objectiveness_weight + ld["loss_box"] * c.box_weight
                + ld["loss_seg"] * c.seg_weight + ld["loss_vote"] * c.vote_weight
                + ld["loss_bc"] * c.bc_weight)
        return loss, ld


def get_model(name):
    return {"BAT": BAT, "P2B": P2B}[name.upper()]
