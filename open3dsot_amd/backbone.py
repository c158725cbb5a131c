"""PointNet++ backbone of the BAT / P2B trackers.

Mirror of models/backbone/pointnet.py::Pointnet_Backbone (:12-88): three single-scale SA
levels (radius 0.3/0.5/0.7, nsample 32, MLPs [C,64,64,128] / [128,128,128,256] /
[256,256,256,256]); only level 0 samples with FPS (when `use_fps`), levels 1-2 keep the
prefix of the previous level's points.  `forward(pointcloud (B,N,3+C), numpoints)` returns
`(xyz_last, feat_last, sample_idxs_level0)` or all levels with `return_intermediate`.
"""
import torch.nn as nn

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
