"""VoteNet-style region proposal head shared by BAT and P2B.

Mirror of models/head/rpn.py::P2BVoteNetRPN (:12-67): per-seed classification MLP, vote MLP
(xyz + feature offsets, residual), vote aggregation = set abstraction (radius 0.3,
nsample 16, MLP [1+f -> +3, v, v, v], no FPS, first `num_proposal` votes as centres),
proposal MLP -> (B, num_proposal, 5) = (x, y, z, theta, objectness).
"""
import torch
from torch import nn

from . import nn_blocks as pt_utils
from .sa_modules import PointnetSAModule


class P2BVoteNetRPN(nn.Module):
    def __init__(self, feature_channel, vote_channel=256, num_proposal=64, normalize_xyz=False):
        super().__init__()
        self.num_proposal = num_proposal
        f = feature_channel
        self.FC_layer_cla = (pt_utils.Seq(f).conv1d(f, bn=True).conv1d(f, bn=True)
                             .conv1d(1, activation=None))
        self.vote_layer = (pt_utils.Seq(3 + f).conv1d(f, bn=True).conv1d(f, bn=True)
                           .conv1d(3 + f, activation=None))
        self.vote_aggregation = PointnetSAModule(
            radius=0.3, nsample=16, mlp=[1 + f, vote_channel, vote_channel, vote_channel],
            use_xyz=True, normalize_xyz=normalize_xyz)
        self.FC_proposal = (pt_utils.Seq(vote_channel).conv1d(vote_channel, bn=True)
                            .conv1d(vote_channel, bn=True).conv1d(3 + 1 + 1, activation=None))

    def forward(self, xyz, feature):
        """xyz (B,N,3), feature (B,f,N) -> boxes (B,P,5), cla (B,N), vote_xyz (B,N,3), centres (B,P,3)"""
        # FC_layer_cla(feature) and vote = seeds + vote_layer(seeds), seeds = cat(xyz^T, feature) (B,3+f,N) (:44-54): two
        # independent stacks over the same seeds, advanced side by side on the GPU (the parts are packed by the stack)
        estimation_cla, vote = pt_utils.seq_apply_pair((self.FC_layer_cla, [feature], False),
                                                       (self.vote_layer, [xyz.transpose(1, 2), feature], True))
        estimation_cla = estimation_cla.squeeze(1)
        score = estimation_cla.sigmoid()
        # split instead of two slices: one backward node (a cat) instead of 2 x (zeros + copy) + add
        v_xyz, v_feat = vote.split([3, vote.shape[1] - 3], dim=1)
        vote_xyz = v_xyz.transpose(1, 2).contiguous()
        vote_feature = torch.cat((score.unsqueeze(1), v_feat), dim=1)
        center_xyzs, proposal_features = self.vote_aggregation(vote_xyz, vote_feature, self.num_proposal)
        offsets = self.FC_proposal(proposal_features)
        o_xyz, o_rest = offsets.split([3, 2], dim=1)
        boxes = torch.cat((o_xyz + center_xyzs.transpose(1, 2), o_rest), dim=1)
        return boxes.transpose(1, 2).contiguous(), estimation_cla, vote_xyz, center_xyzs
