"""VoteNet-style region proposal head shared by BAT and P2B.

Mirror of models/head/rpn.py::P2BVoteNetRPN (:12-67): per-seed classification MLP, vote MLP
(xyz + feature offsets, residual), vote aggregation = set abstraction (radius 0.3,
nsample 16, MLP [1+f -> +3, v, v, v], no FPS, first `num_proposal` votes as centres),
proposal MLP -> (B, num_proposal, 5) = (x, y, z, theta, objectness).
"""
import torch
from torch import nn

import ctypes

from . import capi
from . import nn_blocks as pt_utils
from .sa_modules import PointnetSAModule

_vp, _i, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
capi.register("o3d_rpn_votes_fwd", [_vp, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _vp, _vp, _vp, _vp])
capi.register("o3d_rpn_votes_bwd", [_vp, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _vp, _vp, _vp])
capi.register("o3d_box_assemble", [_vp, _l, _l, _l, _vp, _i, _i, _vp, _vp])

_GLUE = {"on": True}     # TEST hook: False = the torch-op form of the reference (what the kernels are tested against)


def set_fused_glue(enabled):
    _GLUE["on"] = bool(enabled)


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


class RpnVotes(torch.autograd.Function):
    """(estimation_cla (B,N), vote (B,3+f,N)) -> (vote_xyz (B,N,3), vote_feature (B,1+f,N) = cat(sigmoid(cla), vote[:, 3:]))
    -- models/head/rpn.py:47-56 -- one launch each way (csrc/heads.hip::rpn_votes_*)"""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, cla, vote):
        B, C, N = vote.shape
        f = C - 3
        dev = vote.device
        score = torch.empty((B, N), device=dev, dtype=torch.float32)
        vxyz = torch.empty((B, N, 3), device=dev, dtype=torch.float32)
        vfeat = torch.empty((B, 1 + f, N), device=dev, dtype=torch.float32)
        c, v = cla.detach(), vote.detach()
        capi.check(capi.load().o3d_rpn_votes_fwd(c.data_ptr(), c.stride(0), c.stride(1), v.data_ptr(), v.stride(0), v.stride(1),
                                                 v.stride(2), B, N, f, score.data_ptr(), vxyz.data_ptr(), vfeat.data_ptr(),
                                                 _stream(dev)), "rpn_votes_fwd")
        ctx.score = score
        ctx.dims = (B, N, f)
        ctx.set_materialize_grads(False)        # an unused output's gradient stays None (the kernel writes zeros for it)
        return vxyz, vfeat

    @staticmethod
    @capi.on_tensor_device
    def backward(ctx, dvxyz, dvfeat):
        if dvxyz is None and dvfeat is None:
            return None, None
        B, N, f = ctx.dims
        score = ctx.score
        dev = score.device
        dcla = torch.empty((B, N), device=dev, dtype=torch.float32)
        dvote = torch.empty((3 + f, B, N), device=dev, dtype=torch.float32)     # the conv stacks' flat layout
        gx = dvxyz.permute(0, 2, 1) if dvxyz is not None else None               # (B,3,N) view
        capi.check(capi.load().o3d_rpn_votes_bwd(
            score.data_ptr(), dvfeat.data_ptr() if dvfeat is not None else None,
            *(dvfeat.stride() if dvfeat is not None else (0, 0, 0)), gx.data_ptr() if gx is not None else None,
            *(gx.stride() if gx is not None else (0, 0, 0)), B, N, f, dcla.data_ptr(), dvote.data_ptr(), _stream(dev)),
            "rpn_votes_bwd")
        return dcla, dvote.permute(1, 0, 2)


class BoxAssemble(torch.autograd.Function):
    """(offsets (B,5,P), centers (B,P,3)) -> boxes (B,P,5) = [offsets[:, :3]^T + centers, offsets[:, 3:]^T]
    (models/head/rpn.py:62-66); the backward is two views of the incoming gradient"""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, offsets, centers):
        B, _, P = offsets.shape
        o, c = offsets.detach(), centers.detach().contiguous()
        boxes = torch.empty((B, P, 5), device=o.device, dtype=torch.float32)
        capi.check(capi.load().o3d_box_assemble(o.data_ptr(), o.stride(0), o.stride(1), o.stride(2), c.data_ptr(), B, P,
                                                boxes.data_ptr(), _stream(o.device)), "box_assemble")
        return boxes

    @staticmethod
    def backward(ctx, dboxes):
        return dboxes.permute(0, 2, 1), dboxes[:, :, :3]


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
        fused_glue = _GLUE["on"] and vote.is_cuda and vote.dtype == torch.float32 and estimation_cla.dtype == torch.float32
        if fused_glue:       # sigmoid + transposed coordinates + cat(score, features): one launch each way
            vote_xyz, vote_feature = RpnVotes.apply(estimation_cla, vote)
        else:
            score = estimation_cla.sigmoid()
            # split instead of two slices: one backward node (a cat) instead of 2 x (zeros + copy) + add
            v_xyz, v_feat = vote.split([3, vote.shape[1] - 3], dim=1)
            vote_xyz = v_xyz.transpose(1, 2).contiguous()
            vote_feature = torch.cat((score.unsqueeze(1), v_feat), dim=1)
        center_xyzs, proposal_features = self.vote_aggregation(vote_xyz, vote_feature, self.num_proposal)
        offsets = self.FC_proposal(proposal_features)
        if fused_glue and offsets.shape[1] == 5:
            return BoxAssemble.apply(offsets, center_xyzs), estimation_cla, vote_xyz, center_xyzs
        o_xyz, o_rest = offsets.split([3, 2], dim=1)
        boxes = torch.cat((o_xyz + center_xyzs.transpose(1, 2), o_rest), dim=1)
        return boxes.transpose(1, 2).contiguous(), estimation_cla, vote_xyz, center_xyzs
