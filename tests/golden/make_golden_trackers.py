"""Generate tests/golden/ref_trackers.npz by running the REFERENCE'S OWN tracker classes in this container.

Run from the repo root, only where /root/reference exists:  python tests/golden/make_golden_trackers.py
Reference code executed (read-only, from /root/reference): models/bat.py (BAT.__init__, forward :82-112,
compute_loss :57-65, training_step :114-164), models/p2b.py (P2B.__init__, forward :28-59, training_step :61-78),
models/base_model.py (MatchingBaseModel.compute_loss :122-164) and everything they construct
(models/backbone/pointnet.py, models/head/{xcorr,rpn}.py, pointnet2/utils/*).  Stubbed: pytorch_lightning
(`LightningModule` -> nn.Module with no-op save_hyperparameters / log / logger), easydict (attribute dict), nuscenes,
pyquaternion, utils.metrics, datasets.*; `pointnet2_ops._ext` is oracle/ext_shim.py (C restatement, CPU);
`Tensor.cuda()` is the identity.  Weights: tests/golden/det_init.py (closed form, not stored).  Inputs:
open3dsot_amd/synth.py.  Stored: every end point of forward, the loss terms, the weighted loss, the gradients of a
few parameters, the BatchNorm running statistics after the step -- train mode, then eval-mode end points.
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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
torch.Tensor.cuda = lambda self, *a, **k: self
from oracle import ext_shim  # noqa: E402

ext_shim.install()
sys.path.insert(0, REF)
import det_init  # noqa: E402
from open3dsot_amd import synth, trackers  # noqa: E402  (synthetic inputs, the cfg dictionaries)


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


class EasyDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class _Experiment:
    def add_scalars(self, *a, **k):
        pass


class LightningModule(torch.nn.Module):
    global_step = 0

    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, *a, **k):
        pass

    @property
    def logger(self):
        return types.SimpleNamespace(experiment=_Experiment())


stub("pytorch_lightning", LightningModule=LightningModule)
stub("easydict", EasyDict=EasyDict)
stub("nuscenes"); stub("nuscenes.utils", geometry_utils=None); stub("nuscenes.utils.geometry_utils")
stub("pyquaternion", Quaternion=_Dummy)
stub("datasets", points_utils=None); stub("datasets.points_utils"); stub("datasets.data_classes", PointCloud=_Dummy, Box=_Dummy)
stub("utils"); stub("utils.metrics", TorchSuccess=_Dummy, TorchPrecision=_Dummy, estimateOverlap=None, estimateAccuracy=None)
pkg = stub("models"); stub("models.backbone"); stub("models.head")
load("models.backbone.pointnet", "models/backbone/pointnet.py")
load("models.head.xcorr", "models/head/xcorr.py")
load("models.head.rpn", "models/head/rpn.py")
pkg.base_model = load("models.base_model", "models/base_model.py")
ref = {"BAT": load("models.bat", "models/bat.py").BAT, "P2B": load("models.p2b", "models/p2b.py").P2B}
cfgs = {"BAT": trackers.BAT_CAR, "P2B": trackers.P2B_CAR}       # cfgs/BAT_Car.yaml, cfgs/P2B_Car.yaml (model/loss keys)

GRAD_KEYS = ["conv_final.bias", "backbone.SA_modules.0.mlps.0.layer0.conv.weight", "rpn.FC_proposal.2.conv.weight",
             "rpn.vote_layer.0.bn.bn.weight", "xcorr.fea_layer.1.conv.bias"]
out = {}
for name in ("BAT", "P2B"):
    torch.manual_seed(0)
    model = ref[name](EasyDict(cfgs[name]))
    det_init.fill_state_dict(model)
    batch = synth.to_torch(synth.make_batch(40, 2, 256, 512))      # search / 8 >= num_proposal (64)
    captured = {}
    fwd = model.forward

    def rec(b, _f=fwd, _c=captured):
        r = _f(b)
        _c.clear()
        _c.update(r)
        return r
    model.forward = rec
    model.train()
    loss = model.training_step({k: v.clone() for k, v in batch.items()}, 0)
    loss.backward()
    for k, v in captured.items():
        out["%s.train.%s" % (name, k)] = v.detach().numpy().copy()
    out["%s.train.loss" % name] = np.float64(loss.item())
    named = dict(model.named_parameters())
    for k in GRAD_KEYS:
        if k in named:
            out["%s.grad.%s" % (name, k)] = named[k].grad.numpy().copy()
    out["%s.gradnorm" % name] = np.float64(sum(float(p.grad.double().pow(2).sum()) for p in model.parameters() if p.grad is not None) ** 0.5)
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            out["%s.after.%s" % (name, k)] = v.numpy().copy()
    model.eval()
    with torch.no_grad():
        model({k: v.clone() for k, v in batch.items()})
    for k, v in captured.items():
        out["%s.eval.%s" % (name, k)] = v.detach().numpy().copy()
    # the optimizer step the reference would take next (models/base_model.py:28-36: Adam betas (0.5, 0.999), eps 1e-6,
    # StepLR): hyper-parameters and the parameters after ONE step from the gradients above
    conf = model.configure_optimizers()
    opt, sched = conf["optimizer"], conf["lr_scheduler"]
    grp = opt.param_groups[0]
    out["%s.opt.hyper" % name] = np.array([grp["lr"], grp["betas"][0], grp["betas"][1], grp["eps"], grp["weight_decay"],
                                            sched.step_size, sched.gamma], dtype=np.float64)
    opt.step()
    named = dict(model.named_parameters())
    for k in GRAD_KEYS:
        if k in named:
            out["%s.stepped.%s" % (name, k)] = named[k].detach().numpy().copy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_trackers.npz"), **out)
print("wrote ref_trackers.npz:", len(out), "arrays")
