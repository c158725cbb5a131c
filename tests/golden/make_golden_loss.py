"""Generate tests/golden/ref_losses.npz by running the REFERENCE'S OWN loss code in this container.

Run from the repo root, only where /root/reference exists:  python tests/golden/make_golden_loss.py
Reference code executed (read-only, from /root/reference): models/base_model.py
(MatchingBaseModel.compute_loss :122-164) and models/bat.py (BAT.compute_loss :57-65, the loss weighting of
training_step :131-137 is restated below from the same lines).  Stubbed because the packages are absent from the
sandbox (imported at module level, never called on this path): pytorch_lightning (`LightningModule` -> a plain
class), easydict, nuscenes, pyquaternion, shapely-based utils.metrics, datasets.*; `pointnet2_ops._ext` is
oracle/ext_shim.py (bat.py imports the backbone/heads, which import it); `Tensor.cuda()` is the identity
(base_model.py:151 hard-codes it).  The instances are created with `__new__` (no network is built): compute_loss
uses no attribute of `self`.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
torch.Tensor.cuda = lambda self, *a, **k: self
from oracle import ext_shim  # noqa: E402

ext_shim.install()
sys.path.insert(0, REF)


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class _Dummy:
    def __init__(self, *a, **k):
        pass


class LightningModule:      # the reference's BaseModel only needs a base class here
    def __init__(self, *a, **k):
        pass


stub("pytorch_lightning", LightningModule=LightningModule)
stub("easydict", EasyDict=dict)
stub("nuscenes"); stub("nuscenes.utils", geometry_utils=None); stub("nuscenes.utils.geometry_utils")
stub("pyquaternion", Quaternion=_Dummy)
stub("datasets", points_utils=None); stub("datasets.points_utils"); stub("datasets.data_classes", PointCloud=_Dummy, Box=_Dummy)
stub("utils"); stub("utils.metrics", TorchSuccess=_Dummy, TorchPrecision=_Dummy, estimateOverlap=None, estimateAccuracy=None)
pkg = stub("models"); stub("models.backbone"); stub("models.head")
load("models.backbone.pointnet", "models/backbone/pointnet.py")
load("models.head.xcorr", "models/head/xcorr.py")
load("models.head.rpn", "models/head/rpn.py")
base_model = load("models.base_model", "models/base_model.py")
pkg.base_model = base_model
bat = load("models.bat", "models/bat.py")

matching = base_model.MatchingBaseModel.__new__(base_model.MatchingBaseModel)
bat_model = bat.BAT.__new__(bat.BAT)

out = {}
B, N, P, K = 16, 128, 64, 9
for case in range(3):
    g = torch.Generator().manual_seed(100 + case)
    box_label = torch.randn(B, 4, generator=g) * 0.5
    centers = box_label[:, None, :3] + torch.randn(B, P, 3, generator=g) * 0.35
    output = {"estimation_cla": torch.randn(B, N, generator=g) * 2,
              "vote_xyz": box_label[:, None, :3] + torch.randn(B, N, 3, generator=g),
              "estimation_boxes": torch.cat([box_label[:, None, :] + torch.randn(B, P, 4, generator=g),
                                             torch.randn(B, P, 1, generator=g) * 2], 2),
              "center_xyz": centers, "pred_search_bc": torch.randn(B, N, K, generator=g) * 1.5}
    data = {"seg_label": (torch.rand(B, N, generator=g) < 0.3).float(), "box_label": box_label,
            "points2cc_dist_s": torch.randn(B, N, K, generator=g)}
    if case == 2:            # empty denominators: no foreground seed, no proposal within 0.3 m
        data["seg_label"].zero_()
        output["center_xyz"] = centers + 5.0
    grads_in = ["estimation_cla", "vote_xyz", "estimation_boxes", "pred_search_bc"]
    for k in grads_in:
        output[k].requires_grad_(True)
    ld_p2b = matching.compute_loss(data, output)                 # models/base_model.py:122-164
    ld_bat = bat_model.compute_loss(data, output)                # models/bat.py:57-65
    # weighting of models/bat.py:131-137 with cfgs/BAT_Car.yaml:40-44 (objectiveness 1.5, box 0.2, vote 1.0, seg 0.2, bc 1.0)
    total = (ld_bat["loss_objective"] * 1.5 + ld_bat["loss_box"] * 0.2 + ld_bat["loss_seg"] * 0.2
             + ld_bat["loss_vote"] * 1.0 + ld_bat["loss_bc"] * 1.0)
    total.backward()
    for k, v in list(output.items()) + list(data.items()):
        out["c%d.in.%s" % (case, k)] = v.detach().numpy().copy()
    for k, v in ld_bat.items():
        out["c%d.bat.%s" % (case, k)] = np.float64(v.item())
    for k, v in ld_p2b.items():
        out["c%d.p2b.%s" % (case, k)] = np.float64(v.item())
    out["c%d.bat.total" % case] = np.float64(total.item())
    for k in grads_in:
        out["c%d.grad.%s" % (case, k)] = output[k].grad.numpy().copy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_losses.npz"), **out)
print("wrote ref_losses.npz:", len(out), "arrays")
