"""Template <-> search feature fusion heads.

Mirror of models/head/xcorr.py: `P2B_XCorr` :20-53 (cosine similarity (B,M,N) + template
xyz + template features -> SharedMLP [f+4,h,h,h] -> max over the template axis -> two
Conv1d) and `BoxAwareXCorr` :56-103 (k nearest template points in the 9-D BoxCloud space
-> group [xyz | bc | feat] -> SharedMLP [f+12,h,h,h] -> max over k -> two Conv1d).

MI355X differences, results identical up to the documented tie rule:
  * BoxAwareXCorr selects the k nearest by the stable HIP kNN kernel (squared distance,
    ties -> lowest template index) instead of cdist + a full argsort of 64 (:81,:87 --
    torch.argsort's tie order is unspecified);
  * the grouped (B,f+12,N,k) tensor goes through the fused gather+MLP+max path when enabled;
  * P2B_XCorr's (B,f+4,M,N) fusion tensor is never built: its SharedMLP + max over the template axis run on the
    library's kernels with layer 0 split into a per-template-point GEMM and the similarity term
    (open3dsot_amd/fused_xcorr.py, csrc/xcorr.hip).
"""
import torch
from torch import nn
import torch.nn.functional as F

from . import nn_blocks as pt_utils
from . import ops as pointnet2_utils
from . import sa_modules


class BaseXCorr(nn.Module):
    def __init__(self, in_channel, hidden_channel, out_channel):
        super().__init__()
        self.cosine = nn.CosineSimilarity(dim=1)
        self.mlp = pt_utils.SharedMLP([in_channel, hidden_channel, hidden_channel, hidden_channel], bn=True)
        self.fea_layer = (pt_utils.Seq(hidden_channel)
                          .conv1d(hidden_channel, bn=True)
                          .conv1d(out_channel, activation=None))


class P2B_XCorr(BaseXCorr):
    def __init__(self, feature_channel, hidden_channel, out_channel):
        super().__init__(feature_channel + 4, hidden_channel, out_channel)

    def forward(self, template_feature, search_feature, template_xyz):
        """template_feature (B,f,M), search_feature (B,f,N), template_xyz (B,M,3) -> (B,out,N)"""
        B, f, n1 = template_feature.shape
        n2 = search_feature.size(2)
        if sa_modules.fused_enabled() and template_feature.is_cuda:
            from . import fused_xcorr
            if fused_xcorr.supported(self.mlp, template_feature, search_feature):
                # layer 0 split into a per-template-point GEMM + the similarity term (csrc/xcorr.hip): the
                # (B,4+f,M,N) fusion tensor is never built
                fusion = fused_xcorr.p2b_xcorr_mlp_pool(self.mlp, template_feature, search_feature, template_xyz)
                return self.fea_layer(fusion)
        # cosine similarity without materialising the two (B,f,M,N) expansions:
        # sim = <t,s> / (max(|t|,eps) * max(|s|,eps))   (nn.CosineSimilarity, eps=1e-8)
        tn = template_feature.norm(dim=1).clamp_min(1e-8)                      # (B,M)
        sn = search_feature.norm(dim=1).clamp_min(1e-8)                        # (B,N)
        sim = torch.bmm(template_feature.transpose(1, 2), search_feature) / (tn.unsqueeze(2) * sn.unsqueeze(1))
        fusion = torch.cat((sim.unsqueeze(1),
                            template_xyz.transpose(1, 2).unsqueeze(-1).expand(B, 3, n1, n2),
                            template_feature.unsqueeze(-1).expand(B, f, n1, n2)), dim=1)  # (B,1+3+f,M,N)
        fusion = self.mlp(fusion)
        fusion = F.max_pool2d(fusion, kernel_size=[fusion.size(2), 1]).squeeze(2)  # (B,h,N)
        return self.fea_layer(fusion)


class BoxAwareXCorr(BaseXCorr):
    def __init__(self, feature_channel, hidden_channel, out_channel, k=8, use_search_bc=False,
                 use_search_feature=False, bc_channel=9):
        self.k = k
        self.use_search_bc = use_search_bc
        self.use_search_feature = use_search_feature
        c_in = feature_channel + 3 + bc_channel
        if use_search_bc:
            c_in += bc_channel
        if use_search_feature:
            c_in += feature_channel
        super().__init__(c_in, hidden_channel, out_channel)

    def forward(self, template_feature, search_feature, template_xyz, search_xyz=None,
                template_bc=None, search_bc=None):
        """template_* over M points, search_* over N points; bc = (B,*,9) BoxCloud -> (B,out,N)"""
        bundle = torch.cat([template_xyz.transpose(1, 2), template_bc.transpose(1, 2),
                            template_feature], dim=1).contiguous()                 # (B,3+9+f,M)
        # k template points nearest to each search point in BoxCloud space: (B,N,k) int32
        idx = pointnet2_utils.knn_point(self.k, search_bc.detach(), template_bc.detach())
        extra = []
        if self.use_search_feature:
            extra.append(search_feature.unsqueeze(-1).expand(-1, -1, -1, self.k))
        if self.use_search_bc:
            extra.append(search_bc.transpose(1, 2).unsqueeze(-1).expand(-1, -1, -1, self.k))
        if not extra and sa_modules.fused_enabled() and bundle.is_cuda:
            from . import fused
            if fused.supports_mlp(self.mlp):
                fusion = fused.group_mlp_pool(self.mlp, bundle, idx)                # (B,h,N)
                return self.fea_layer(fusion)
        grouped = pointnet2_utils.grouping_operation(bundle, idx)                  # (B,3+9+f,N,k)
        if extra:
            grouped = torch.cat(extra + [grouped], dim=1)
        fusion = self.mlp(grouped)
        fusion, _ = torch.max(fusion, dim=-1)
        return self.fea_layer(fusion)
