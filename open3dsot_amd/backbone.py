"""PointNet++ backbone of the BAT / P2B trackers.

Mirror of models/backbone/pointnet.py::Pointnet_Backbone (:12-88): three single-scale SA
levels (radius 0.3/0.5/0.7, nsample 32, MLPs [C,64,64,128] / [128,128,128,256] /
[256,256,256,256]); only level 0 samples with FPS (when `use_fps`), levels 1-2 keep the
prefix of the previous level's points.  `forward(pointcloud (B,N,3+C), numpoints)` returns
`(xyz_last, feat_last, sample_idxs_level0)` or all levels with `return_intermediate`.
"""
import torch
import torch.nn as nn

from . import nn_blocks
from .sa_modules import PointnetSAModule

_LEVELS = ((0.3, (64, 64, 128)), (0.5, (128, 128, 256)), (0.7, (256, 256, 256)))


class Pointnet_Backbone(nn.Module):
    def __init__(self, use_fps=False, normalize_xyz=False, return_intermediate=False, input_channels=0):
        super().__init__()
        self.return_intermediate = return_intermediate
        self.SA_modules = nn.ModuleList()
        c_in = input_channels
        for level, (radius, widths) in enumerate(_LEVELS):
            self.SA_modules.append(PointnetSAModule(
                radius=radius, nsample=32, mlp=[c_in] + list(widths), use_xyz=True,
                use_fps=use_fps and level == 0, normalize_xyz=normalize_xyz))
            c_in = widths[-1]

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    def forward(self, pointcloud, numpoints):
        xyz, features = self._break_up_pc(pointcloud)
        l_xyz, l_features, idx0 = [xyz], [features], None
        for i, sa in enumerate(self.SA_modules):
            nxyz, nfeat, sidx = sa(l_xyz[i], l_features[i], numpoints[i], True)
            l_xyz.append(nxyz)
            l_features.append(nfeat)
            if i == 0:
                idx0 = sidx
        if self.return_intermediate:
            return l_xyz[1:], l_features[1:], idx0
        return l_xyz[-1], l_features[-1], idx0


def _backbone_forward_pair(self, pc_a, numpoints_a, pc_b, numpoints_b, sample_idxs=None, geometry=None):
    """self(pc_a, numpoints_a), self(pc_b, numpoints_b) with every level's two module calls issued as one
    (sa_modules.forward_pair): same numbers as the two calls in that order.  sample_idxs = (idx_a, idx_b): level 0's
    farthest-point-sampling indices when they were computed ahead of the step (`sampling_indices`).  geometry: one dict per
    level (`pair_geometry`) when the levels' coordinate-only parts were computed ahead of the step as well."""
    xyz_a, feat_a = self._break_up_pc(pc_a)
    xyz_b, feat_b = self._break_up_pc(pc_b)
    la, lb = ([xyz_a], [feat_a]), ([xyz_b], [feat_b])
    idx0 = [None, None]
    for i, sa in enumerate(self.SA_modules):
        ra, rb = sa.forward_pair(la[0][i], la[1][i], numpoints_a[i], lb[0][i], lb[1][i], numpoints_b[i],
                                 sample_idxs if i == 0 else None, geo=geometry[i] if geometry is not None else None)
        for k, (lst, r) in enumerate(((la, ra), (lb, rb))):
            lst[0].append(r[0])
            lst[1].append(r[1])
            if i == 0:
                idx0[k] = r[2]
    if self.return_intermediate:
        return (la[0][1:], la[1][1:], idx0[0]), (lb[0][1:], lb[1][1:], idx0[1])
    return (la[0][-1], la[1][-1], idx0[0]), (lb[0][-1], lb[1][-1], idx0[1])


Pointnet_Backbone.forward_pair = _backbone_forward_pair


def sampling_indices(self, pc_a, npoint_a, pc_b, npoint_b):
    """level 0's sampling indices of both clouds -- the part of the backbone that depends on the input clouds only
    (pointnet2_modules.py:52-56) -- or None when level 0 does not sample with FPS.  One launch for both clouds."""
    if not self.SA_modules[0].use_fps:
        return None
    from . import ops
    return ops.furthest_point_sample_pair(pc_a[..., 0:3].contiguous(), npoint_a, pc_b[..., 0:3].contiguous(), npoint_b)


Pointnet_Backbone.sampling_indices = sampling_indices


def pair_geometry(self, pc_a, numpoints_a, pc_b, numpoints_b, sample_idxs=None, dst=None):
    """[dict per level] -- for every level of forward_pair the part that depends on the input COORDINATES only: level 0's
    centres are the (farthest-point or prefix) samples of the clouds, the next level's clouds are those centres, and so on
    (models/backbone/pointnet.py:66-88 with pointnet2_modules.py:52-62): no feature enters.  None when some level would
    not take the fused paired path (the step then computes everything inline).  dst: one dict of destination tensors per level."""
    xyz_a, xyz_b = pc_a[..., 0:3].contiguous(), pc_b[..., 0:3].contiguous()
    out = []
    for i, sa in enumerate(self.SA_modules):
        geo = sa.pair_geometry(xyz_a, numpoints_a[i], xyz_b, numpoints_b[i], sample_idxs if i == 0 else None,
                               out=dst[i] if dst is not None else None)
        if geo is None:
            return None
        out.append(geo)
        B, na, nb = xyz_a.shape[0], numpoints_a[i], numpoints_b[i]
        xyz_a = geo["centers"][:B * na].view(B, na, 3)
        xyz_b = geo["centers"][B * na:B * (na + nb)].view(B, nb, 3)
    return out


Pointnet_Backbone.pair_geometry = pair_geometry


def _pointwise_chain(x, layers, pool=False):
    """(Conv1d k=1 -> BatchNorm1d -> ReLU)* on (B,C,N) -- the M2-Track pointwise stack is the grouped MLP
    with one ball per cloud.  `layers` = [(conv, bn, relu)]; returns (B,C',N), or
    the global max over N (B,C') with `pool`.  On the GPU it runs on the library's fp32-MFMA GEMM kernels
    (open3dsot_amd/fused_pointwise.py); otherwise as GEMMs on the flat (C, B*N) layout with torch ops
    (same statistics as BatchNorm1d on (B,C,N))."""
    if x.is_cuda and _FUSED_PW["on"]:
        from . import fused_pointwise
        pairs = [(conv, bn) for conv, bn, _ in layers]
        if fused_pointwise.supported(x, pairs):
            return fused_pointwise.chain(x, pairs, "gmax" if pool else "act")       # (any strides: a stack's output is a (B,C,N) view of its flat (C, B*N) buffer and goes into the next stack as it is)
    h = _pointwise_chain_torch(x, layers)
    return h.amax(dim=2) if pool else h


def _cloud_chain(second, pooled, layers):
    """(Conv1d -> BatchNorm1d -> ReLU)* on cat([second, pooled broadcast over the points]) (models/backbone/pointnet.py:
    188-193); on the GPU the broadcast block becomes a per-cloud bias (fused_pointwise.chain_cloud)"""
    if second.is_cuda and _FUSED_PW["on"] and _FUSED_PW["cloud_bias"]:
        from . import fused_pointwise
        pairs = [(conv, bn) for conv, bn, _ in layers]
        if fused_pointwise.cloud_supported(second, pooled, pairs):
            return fused_pointwise.chain_cloud(second, pooled, pairs)
    x = torch.cat([second, pooled.unsqueeze(-1).expand(-1, -1, second.shape[2])], dim=1)
    return _pointwise_chain(x, layers)


_FUSED_PW = {"on": True, "cloud_bias": True}


def set_cloud_bias(enabled):
    """SegPointNet's pooled-feature block as a per-cloud bias (default) or as 1024 broadcast GEMM rows (A/B, tests)"""
    _FUSED_PW["cloud_bias"] = bool(enabled)


def set_fused_pointwise(enabled):
    _FUSED_PW["on"] = bool(enabled)


def _pointwise_chain_torch(x, layers):
    B, C, N = x.shape
    h = x.permute(1, 0, 2).reshape(C, B * N)
    for conv, bn, act in layers:
        h = torch.mm(conv.weight[:, :, 0], h)
        if conv.bias is not None:
            h = h + conv.bias[:, None]
        h = act(bn(h.unsqueeze(0)).squeeze(0))
    return h.reshape(-1, B, N).permute(1, 0, 2)


def _triples(mods):
    mods = list(mods)
    return [(mods[i], mods[i + 1], mods[i + 2]) for i in range(0, len(mods), 3)]


class MiniPointNet(nn.Module):
    """Per-point MLP -> global max -> hidden MLP (-> fc).  Mirror of models/backbone/pointnet.py:91-141;
    the module tree (`features.<i>`, `fc`) and therefore the state_dict keys equal the reference's."""

    def __init__(self, input_channel, per_point_mlp, hidden_mlp, output_size=0):
        super().__init__()
        seq, c = [], input_channel
        for width in per_point_mlp:
            seq += [nn.Conv1d(c, width, 1), nn.BatchNorm1d(width), nn.ReLU()]
            c = width
        self._n_point = len(seq)
        seq += [nn.AdaptiveMaxPool1d(output_size=1), nn.Flatten()]
        for width in hidden_mlp:
            seq += [nn.Linear(c, width), nn_blocks.RowBatchNorm1d(width), nn.ReLU()]
            c = width
        self.features = nn.Sequential(*seq)
        self.output_size = output_size
        if output_size >= 0:
            self.fc = nn.Linear(c, output_size)

    def forward(self, x):
        """x (B,C,N) -> (B, hidden_mlp[-1]) or (B, output_size)"""
        if nn_blocks._FLAT["on"]:
            mods = list(self.features)
            x = _pointwise_chain(x, _triples(mods[:self._n_point]), pool=True)
            from . import fused_rows
            x = fused_rows.seq_rows(mods[self._n_point + 2:], x)     # Linear -> BatchNorm1d -> ReLU rows: one launch each
        else:
            x = self.features(x)
        return self.fc(x) if self.output_size > 0 else x


class SegPointNet(nn.Module):
    """Per-point MLP, global max appended to the second layer's features, second per-point MLP, 1x1
    conv head.  Mirror of models/backbone/pointnet.py:144-204 (same `seq_per_point.<i>.<j>` keys)."""

    def __init__(self, input_channel, per_point_mlp1, per_point_mlp2, output_size=0, return_intermediate=False):
        super().__init__()
        self.return_intermediate = return_intermediate
        self.seq_per_point = nn.ModuleList()
        c = input_channel
        for width in per_point_mlp1:
            self.seq_per_point.append(nn.Sequential(nn.Conv1d(c, width, 1), nn.BatchNorm1d(width), nn.ReLU()))
            c = width
        self.pool = nn.AdaptiveMaxPool1d(output_size=1)
        self.seq_per_point2 = nn.ModuleList()
        c = c + per_point_mlp1[1]
        for width in per_point_mlp2:
            self.seq_per_point2.append(nn.Sequential(nn.Conv1d(c, width, 1), nn.BatchNorm1d(width), nn.ReLU()))
            c = width
        self.output_size = output_size
        if output_size >= 0:
            self.fc = nn.Conv1d(c, output_size, 1)

    def forward(self, x):
        """x (B,C,N) -> (B,output_size,N) [, pooled (B,C1)]"""
        if nn_blocks._FLAT["on"]:
            first = [tuple(m) for m in self.seq_per_point]
            second = _pointwise_chain(x, first[:2])
            pooled = _pointwise_chain(second, first[2:], pool=True).unsqueeze(-1)      # (B,C1,1)
            x = _cloud_chain(second, pooled.squeeze(-1), [tuple(m) for m in self.seq_per_point2])
            if self.output_size > 0:
                x = nn_blocks.pointwise_conv1d(self.fc, x)          # (x: a view of the stack's flat buffer, consumed in place)
        else:
            second = None
            for i, m in enumerate(self.seq_per_point):
                x = m(x)
                if i == 1:
                    second = x
            pooled = self.pool(x)
            x = torch.cat([second, pooled.expand_as(x)], dim=1)
            for m in self.seq_per_point2:
                x = m(x)
            if self.output_size > 0:
                x = self.fc(x)
        if self.return_intermediate:
            return x, pooled.squeeze(-1)
        return x
