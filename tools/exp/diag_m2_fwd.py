#!/usr/bin/env python
"""GPU diagnostic (round 6): M2-Track training FORWARD at 48 x 2 048 points against the host mirror in fp64 (CPU), with the
per-cloud bias of SegPointNet's second stack on and off: outputs and the running statistics of every BatchNorm."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import test_golden_m2track as T
from open3dsot_amd import backbone, m2track, nn_blocks

gold = np.load(os.path.join(ROOT, "tests/golden/ref_m2track.npz"))
gold48 = np.load(os.path.join(ROOT, "tests/golden/ref_m2track_b48.npz"))
goldg = np.load(os.path.join(ROOT, "tests/golden/ref_m2track_grad.npz"))
tag = sys.argv[1] if len(sys.argv) > 1 else "b48x2048"


def build(dev, dtype):
    net = m2track.M2TRACK()
    net.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd.")}, strict=True)
    return net.to(dev).to(dtype).train()


def fwd(dev, dtype, flat=True):
    nn_blocks.set_flat_pointwise(flat)
    net = build(dev, dtype)
    b = {k: (v.to(dev).to(dtype) if v.dtype.is_floating_point else v.to(dev)) for k, v in T.grad_fixture_batch(tag, gold, gold48, goldg).items()}
    with torch.no_grad(), T.replay_hard_masks(goldg, tag):
        out = net(b)
    nn_blocks.set_flat_pointwise(True)
    return {k: v.detach().double().cpu() for k, v in out.items()}, {k: v.detach().double().cpu() for k, v in net.state_dict().items() if "running" in k}


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


o64, s64 = fwd("cpu", torch.float64, flat=False)
o32, s32 = fwd("cpu", torch.float32, flat=False)
res = {}
for cb in (True, False):
    backbone.set_cloud_bias(cb)
    res[cb] = fwd("cuda", torch.float32)
backbone.set_cloud_bias(True)
print("%s  quantity: GPU cloud-bias ON | GPU cloud-bias OFF | CPU fp32 module path   (max abs err / max abs vs the fp64 CPU mirror)" % tag)
for k in o64:
    print("  out.%-28s %.2e | %.2e | %.2e" % (k, rel(res[True][0][k], o64[k]), rel(res[False][0][k], o64[k]), rel(o32[k], o64[k])))
for k in s64:
    a, b, c = rel(res[True][1][k], s64[k]), rel(res[False][1][k], s64[k]), rel(s32[k], s64[k])
    if max(a, b) > 3e-6:
        print("  %-32s %.2e | %.2e | %.2e" % (k, a, b, c))
