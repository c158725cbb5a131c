"""Import-level stubs that let the reference's tracker classes (models/bat.py::BAT, models/p2b.py::P2B, on
models/base_model.py::MatchingBaseModel) be imported without pytorch_lightning, easydict, nuscenes, pyquaternion,
torchmetrics ...: `LightningModule` becomes an nn.Module with no-op save_hyperparameters / log / logger, `EasyDict` an
attribute dict, the dataset / metric modules empty.  Nothing of the reference's arithmetic is replaced.  Used by the
fixture generators (tests/golden/make_golden_trackers*.py) and by tests/test_reference_modules_gpu.py.  TEST ONLY."""
import importlib.util
import os
import sys
import types

import torch


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


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_trackers(ref_root):
    """-> {"BAT": reference BAT class, "P2B": reference P2B class}; `pointnet2_ops._ext` must already be importable
    (the oracle shim on the CPU, this repo's HIP drop-in on the GPU) and `ref_root` on sys.path"""
    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ref_root, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
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
    return {"BAT": load("models.bat", "models/bat.py").BAT, "P2B": load("models.p2b", "models/p2b.py").P2B}
