"""Generate tests/golden/ref_cold_api.npz from the REFERENCE'S OWN cold operator API:
`PointnetFPModule` (pointnet2/utils/pointnet2_modules.py:152-212), `GroupAll` (pointnet2_utils.py:342-385) and
`knn_point` (:388-402, torch.cdist + argsort -- also what BoxAwareXCorr does at models/head/xcorr.py:81-87).

Run from the repo root, only where /root/reference exists (the build container):
    python tests/golden/make_golden_cold.py
`pointnet2_ops._ext` is the oracle shim (oracle/ext_shim.py), as in make_golden.py.  The kNN cases record the
reference's selection on (a) continuous random BoxClouds (no ties) and (b) BoxClouds quantised to a 0.25 m grid
(many EXACT ties, which torch.argsort orders arbitrarily and cdist's matmul path perturbs by rounding): the tests
require identical neighbours on (a) and identical neighbour DISTANCES on (b) -- the documented divergence.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import ext_shim  # noqa: E402

ext_shim.install()
sys.path.insert(0, REF)
from pointnet2.utils import pointnet2_modules as ref_modules  # noqa: E402
from pointnet2.utils import pointnet2_utils as ref_utils  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    g = torch.Generator().manual_seed(8814)
    torch.manual_seed(8814)
    fix = {}
    # ---- PointnetFPModule: 3-NN interpolation of 24 known points onto 40 unknown points + SharedMLP [20+12, 32, 16]
    B, n, m, C1, C2 = 2, 40, 24, 12, 20
    unknown = torch.randn(B, n, 3, generator=g)
    known = torch.cat([unknown[:, :8] + 0.0, torch.randn(B, m - 8, 3, generator=g)], 1).contiguous()   # 8 coincident points
    uf = torch.randn(B, C1, n, generator=g)
    kf = torch.randn(B, C2, m, generator=g)
    fp = ref_modules.PointnetFPModule([C1 + C2, 32, 16], bn=True)
    for mod in fp.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5, generator=g)
            mod.bias.data.normal_(0, 0.1, generator=g)
            mod.running_mean.data.normal_(0, 0.1, generator=g)
            mod.running_var.data.uniform_(0.5, 1.5, generator=g)
    for k, v in fp.state_dict().items():
        fix["fp.sd." + k] = v.detach().numpy().copy()
    for nm, t in (("unknown", unknown), ("known", known), ("unknow_feats", uf), ("known_feats", kf)):
        fix["fp.in." + nm] = t.numpy()
    import copy
    for mode in ("train", "eval"):
        mod = copy.deepcopy(fp).train(mode == "train")
        a, b = uf.clone().requires_grad_(True), kf.clone().requires_grad_(True)
        out = mod(unknown, known, a, b)
        ct = torch.randn(out.shape, generator=torch.Generator().manual_seed(3))
        (out * ct).sum().backward()
        fix["fp.%s.out" % mode] = out.detach().numpy()
        fix["fp.%s.d_unknow_feats" % mode] = a.grad.numpy()
        fix["fp.%s.d_known_feats" % mode] = b.grad.numpy()
        if mode == "train":
            fix["fp.ct"] = ct.numpy()
            for k, v in mod.state_dict().items():
                if "running" in k:
                    fix["fp.train.sd_after." + k] = v.detach().numpy().copy()
    # no unknown features: interpolation only feeds the MLP
    fp2 = ref_modules.PointnetFPModule([C2, 8], bn=True).eval()
    for k, v in fp2.state_dict().items():
        fix["fp2.sd." + k] = v.detach().numpy().copy()
    fix["fp2.eval.out"] = fp2(unknown, known, None, kf).detach().numpy()

    # ---- GroupAll
    xyz = torch.randn(B, 10, 3, generator=g)
    feats = torch.randn(B, 5, 10, generator=g)
    fix["ga.in.xyz"], fix["ga.in.feats"] = xyz.numpy(), feats.numpy()
    fix["ga.xyz_feats"] = ref_utils.GroupAll(use_xyz=True)(xyz, None, feats).numpy()
    fix["ga.feats_only"] = ref_utils.GroupAll(use_xyz=False)(xyz, None, feats).numpy()
    fix["ga.xyz_only"] = ref_utils.GroupAll(use_xyz=True)(xyz, None, None).numpy()

    # ---- knn_point: continuous and gridded 9-D BoxClouds, 128 queries x 64 references, k = 4 (BAT_Car.yaml:34)
    q = torch.rand(3, 128, 9, generator=g) * 3
    r = torch.rand(3, 64, 9, generator=g) * 3
    fix["knn.cont.q"], fix["knn.cont.r"] = q.numpy(), r.numpy()
    fix["knn.cont.idx"] = ref_utils.knn_point(4, q, r).numpy()
    qg, rg = torch.round(q * 4) / 4, torch.round(r[:, :, :] * 4) / 4
    rg[:, 32:] = rg[:, :32]                  # every reference point has an exact duplicate
    fix["knn.grid.q"], fix["knn.grid.r"] = qg.numpy(), rg.numpy()
    fix["knn.grid.idx"] = ref_utils.knn_point(4, qg, rg).numpy()

    path = os.path.join(OUT, "ref_cold_api.npz")
    np.savez_compressed(path, **fix)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1e3), len(fix), "arrays")


if __name__ == "__main__":
    main()
