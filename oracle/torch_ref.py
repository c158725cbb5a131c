"""Pure-PyTorch fp32 CPU restatement of the BAT / P2B hot path (TEST INFRASTRUCTURE).

An INDEPENDENT functional composition (F.conv / F.batch_norm on a flat state_dict with the
reference's key names) of what these reference files compute:
  pointnet2/utils/pointnet2_utils.py:299-339   QueryAndGroup
  pointnet2/utils/pointnet2_modules.py:31-79   set abstraction (sample, group, MLP, max)
  pointnet2/utils/pytorch_utils.py:12-37,68-121 SharedMLP / Conv(+BN)(+ReLU) order and bias rule
  models/backbone/pointnet.py:28-88            three SA levels
  models/head/xcorr.py:20-103                  P2B_XCorr / BoxAwareXCorr
  models/head/rpn.py:12-67                     P2BVoteNetRPN
  models/bat.py:57-65,82-143  models/p2b.py:39-78  models/base_model.py:122-164
Index operators come from the plain-C oracle (oracle/ops.py); gathers are written with
torch.gather so autograd supplies a deterministic backward.  It is pinned against the
reference's own modules by tests/golden (generated in the build container by importing the
reference files, see tests/golden/make_golden.py).  This is also the timed `cpu_baseline`
("port") of bench.py -- the reference has no CPU path of its own.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops

EPS, MOMENTUM = 1e-5, 0.1


def _np(t):
    return t.detach().cpu().float().contiguous().numpy()   # index ops are fp32 (also under an fp64 shadow run)


def fps(xyz, npoint):
    return torch.from_numpy(ops.furthest_point_sampling(_np(xyz), npoint))


def ball(xyz, new_xyz, radius, nsample):
    return torch.from_numpy(ops.ball_query(_np(new_xyz), _np(xyz), float(np.float32(radius)), nsample))


def knn(query, ref, k):
    return torch.from_numpy(ops.knn(_np(query), _np(ref), k))


def group(feats, idx):
    """feats (B,C,N), idx (B,np,ns) -> (B,C,np,ns) through a differentiable gather."""
    B, C, _ = feats.shape
    _, n_p, n_s = idx.shape
    flat = idx.long().reshape(B, 1, n_p * n_s).expand(B, C, n_p * n_s)
    return feats.gather(2, flat).reshape(B, C, n_p, n_s)


class State:
    """state_dict view with train/eval BatchNorm semantics (running stats updated in place)."""

    def __init__(self, sd, training):
        self.sd, self.training = sd, training

    def conv_bn_relu(self, prefix, x, dims, relu=True):
        """<prefix>.conv (+ <prefix>.bn.bn) (+ ReLU); dims = 1 or 2."""
        w = self.sd[prefix + ".conv.weight"]
        b = self.sd.get(prefix + ".conv.bias")
        x = F.conv1d(x, w, b) if dims == 1 else F.conv2d(x, w, b)
        if prefix + ".bn.bn.weight" in self.sd:
            rm, rv = self.sd[prefix + ".bn.bn.running_mean"], self.sd[prefix + ".bn.bn.running_var"]
            x = F.batch_norm(x, rm, rv, self.sd[prefix + ".bn.bn.weight"], self.sd[prefix + ".bn.bn.bias"],
                             self.training, MOMENTUM, EPS)
            if self.training:
                self.sd[prefix + ".bn.bn.num_batches_tracked"] += 1
        return F.relu(x) if relu else x

    def shared_mlp(self, prefix, x, nlayers=3):
        for j in range(nlayers):
            x = self.conv_bn_relu("%s.layer%d" % (prefix, j), x, 2)
        return x

    def seq(self, prefix, x, n):
        """pt_utils.Seq of n Conv1d blocks; the last one has no bn / activation in every caller."""
        for j in range(n):
            x = self.conv_bn_relu("%s.%d" % (prefix, j), x, 1, relu=(j < n - 1))
        return x


def set_abstraction(st, prefix, xyz, feats, npoint, radius, nsample, use_fps):
    """-> new_xyz (B,npoint,3), feats (B,C',npoint), sample_idxs (B,npoint) i32"""
    B = xyz.shape[0]
    if use_fps:
        sidx = fps(xyz, npoint)
    else:
        sidx = torch.arange(npoint, dtype=torch.int32).repeat(B, 1)
    new_xyz = xyz.gather(1, sidx.long()[:, :, None].expand(B, npoint, 3))
    idx = ball(xyz, new_xyz, radius, nsample)
    g_xyz = group(xyz.transpose(1, 2), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
    x = g_xyz if feats is None else torch.cat([g_xyz, group(feats, idx)], dim=1)
    x = st.shared_mlp(prefix + ".mlps.0", x)
    return new_xyz, x.max(dim=3)[0], sidx


def backbone(st, pc, numpoints, use_fps):
    xyz, feats = pc[..., :3].contiguous(), None
    idx0 = None
    for i, (radius, npnt) in enumerate(zip((0.3, 0.5, 0.7), numpoints)):
        xyz, feats, sidx = set_abstraction(st, "backbone.SA_modules.%d" % i, xyz, feats, npnt, radius, 32,
                                           use_fps and i == 0)
        if i == 0:
            idx0 = sidx
    return xyz, feats, idx0


def rpn(st, xyz, feature, num_proposal=64):
    cla = st.seq("rpn.FC_layer_cla", feature, 3).squeeze(1)
    score = torch.sigmoid(cla)
    seeds = torch.cat((xyz.transpose(1, 2), feature), dim=1)
    vote = seeds + st.seq("rpn.vote_layer", seeds, 3)
    vote_xyz = vote[:, 0:3, :].transpose(1, 2).contiguous()
    vote_feature = torch.cat((score.unsqueeze(1), vote[:, 3:, :]), dim=1)
    centers, prop, _ = set_abstraction(st, "rpn.vote_aggregation", vote_xyz, vote_feature, num_proposal, 0.3, 16, False)
    off = st.seq("rpn.FC_proposal", prop, 3)
    boxes = torch.cat((off[:, 0:3, :] + centers.transpose(1, 2), off[:, 3:5, :]), dim=1).transpose(1, 2).contiguous()
    return boxes, cla, vote_xyz, centers


def box_aware_xcorr(st, t_feat, s_feat, t_xyz, t_bc, s_bc, k):
    bundle = torch.cat([t_xyz.transpose(1, 2), t_bc.transpose(1, 2), t_feat], dim=1)
    idx = knn(s_bc, t_bc, k)                              # (B,N,k): stable, lowest index first
    x = st.shared_mlp("xcorr.mlp", group(bundle, idx))
    return st.seq("xcorr.fea_layer", x.max(dim=-1)[0], 2)


def p2b_xcorr(st, t_feat, s_feat, t_xyz):
    B, f, n1 = t_feat.shape
    n2 = s_feat.shape[2]
    sim = F.cosine_similarity(t_feat.unsqueeze(-1).expand(B, f, n1, n2), s_feat.unsqueeze(2).expand(B, f, n1, n2), dim=1)
    x = torch.cat((sim.unsqueeze(1), t_xyz.transpose(1, 2).unsqueeze(-1).expand(B, 3, n1, n2),
                   t_feat.unsqueeze(-1).expand(B, f, n1, n2)), dim=1)
    x = st.shared_mlp("xcorr.mlp", x)
    return st.seq("xcorr.fea_layer", x.max(dim=2)[0], 2)


def bat_forward(sd, batch, training, k=4, bc_channel=9, num_proposal=64, use_fps=True):
    st = State(sd, training)
    template, search, t_bc = batch["template_points"], batch["search_points"], batch["points2cc_dist_t"]
    M, N = template.shape[1], search.shape[1]
    t_xyz, t_feat, t_idx = backbone(st, template, [M // 2, M // 4, M // 8], use_fps)
    s_xyz, s_feat, s_idx = backbone(st, search, [N // 2, N // 4, N // 8], use_fps)
    t_feat = F.conv1d(t_feat, sd["conv_final.weight"], sd["conv_final.bias"])
    s_feat = F.conv1d(s_feat, sd["conv_final.weight"], sd["conv_final.bias"])
    pred_bc = st.seq("mlp_bc", torch.cat([s_xyz.transpose(1, 2), s_feat], dim=1), 3).transpose(1, 2)
    t_bc = t_bc.gather(1, t_idx[:, :M // 8, None].long().expand(-1, -1, bc_channel))
    fusion = box_aware_xcorr(st, t_feat, s_feat, t_xyz, t_bc, pred_bc, k)
    boxes, cla, vote_xyz, centers = rpn(st, s_xyz, fusion, num_proposal)
    return {"estimation_boxes": boxes, "estimation_cla": cla, "vote_xyz": vote_xyz, "center_xyz": centers,
            "sample_idxs": s_idx, "pred_search_bc": pred_bc}


def p2b_forward(sd, batch, training, num_proposal=64, use_fps=False):
    st = State(sd, training)
    template, search = batch["template_points"], batch["search_points"]
    M, N = template.shape[1], search.shape[1]
    t_xyz, t_feat, _ = backbone(st, template, [M // 2, M // 4, M // 8], use_fps)
    s_xyz, s_feat, s_idx = backbone(st, search, [N // 2, N // 4, N // 8], use_fps)
    t_feat = F.conv1d(t_feat, sd["conv_final.weight"], sd["conv_final.bias"])
    s_feat = F.conv1d(s_feat, sd["conv_final.weight"], sd["conv_final.bias"])
    fusion = p2b_xcorr(st, t_feat, s_feat, t_xyz)
    boxes, cla, vote_xyz, centers = rpn(st, s_xyz, fusion, num_proposal)
    return {"estimation_boxes": boxes, "estimation_cla": cla, "vote_xyz": vote_xyz, "center_xyz": centers,
            "sample_idxs": s_idx}


def matching_loss(batch, out, weights, bat=True):
    """Weighted training loss incl. the label re-indexing of training_step; returns (loss, dict)."""
    n_seed = out["estimation_cla"].shape[1]
    sidx = out["sample_idxs"][:, :n_seed].long()
    seg = batch["seg_label"].gather(1, sidx)
    box_label = batch["box_label"]
    boxes, cla, centers, vote_xyz = out["estimation_boxes"], out["estimation_cla"], out["center_xyz"], out["vote_xyz"]
    l_seg = F.binary_cross_entropy_with_logits(cla, seg)
    l_vote = F.smooth_l1_loss(vote_xyz, box_label[:, None, :3].expand_as(vote_xyz), reduction="none")
    l_vote = (l_vote.mean(2) * seg).sum() / (seg.sum() + 1e-06)
    dist = torch.sqrt(torch.sum((centers - box_label[:, None, :3]) ** 2, dim=-1) + 1e-6)
    label = torch.zeros_like(dist)
    label[dist < 0.3] = 1
    mask = torch.zeros_like(dist)
    mask[dist < 0.3] = 1
    mask[dist > 0.6] = 1
    l_obj = F.binary_cross_entropy_with_logits(boxes[:, :, 4], label, pos_weight=dist.new_tensor([2.0]))
    l_obj = torch.sum(l_obj * mask) / (torch.sum(mask) + 1e-6)
    l_box = F.smooth_l1_loss(boxes[:, :, :4], box_label[:, None, :4].expand_as(boxes[:, :, :4]), reduction="none")
    l_box = torch.sum(l_box.mean(2) * label) / (label.sum() + 1e-6)
    ld = {"loss_objective": l_obj, "loss_box": l_box, "loss_seg": l_seg, "loss_vote": l_vote}
    loss = (l_obj * weights["objectiveness_weight"] + l_box * weights["box_weight"]
            + l_seg * weights["seg_weight"] + l_vote * weights["vote_weight"])
    if bat:
        s_bc = batch["points2cc_dist_s"].gather(1, sidx[:, :, None].expand(-1, -1, out["pred_search_bc"].shape[2]))
        l_bc = F.smooth_l1_loss(out["pred_search_bc"], s_bc, reduction="none")
        l_bc = torch.sum(l_bc.mean(2) * seg) / (seg.sum() + 1e-6)
        ld["loss_bc"] = l_bc
        loss = loss + l_bc * weights["bc_weight"]
    return loss, ld
