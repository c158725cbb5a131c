#!/usr/bin/env python
"""GPU diagnostic: the second-stage head's input X = mini_pointnet2(...) and dL/dX, cloud bias on vs off (48 x 2 048)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import test_golden_m2track as T
from open3dsot_amd import backbone, fused_rows, m2track

gold = np.load(os.path.join(ROOT, "tests/golden/ref_m2track.npz"))
gold48 = np.load(os.path.join(ROOT, "tests/golden/ref_m2track_b48.npz"))
goldg = np.load(os.path.join(ROOT, "tests/golden/ref_m2track_grad.npz"))
tag = "b48x2048"
fused_rows.set_fused_rows(False)
cap = {}
for cb in (True, False):
    backbone.set_cloud_bias(cb)
    net = m2track.M2TRACK()
    net.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd.")}, strict=True)
    net = net.cuda().train()
    c = cap[cb] = {}

    def hook(name):
        def f(mod, inp, out):
            x = inp[0]
            c[name + ".in"] = x
            if x.requires_grad:
                x.retain_grad()
            c[name + ".out"] = out
            out.retain_grad()
        return f
    net.box_mlp[0].register_forward_hook(hook("box_mlp.0"))
    net.box_mlp[1].register_forward_hook(hook("box_mlp.1"))
    net.box_mlp[3].register_forward_hook(hook("box_mlp.3"))
    b = {k: v.cuda() for k, v in T.grad_fixture_batch(tag, gold, gold48, goldg).items()}
    with T.replay_hard_masks(goldg, tag):
        out = net(b)
        ld = net.compute_loss(b, out)
    ld["loss_total"].backward()
    c["grads"] = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
backbone.set_cloud_bias(True)


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


on, off = cap[True], cap[False]
for k in ("box_mlp.0.in", "box_mlp.0.out", "box_mlp.1.out", "box_mlp.3.out"):
    print("%-16s value ON vs OFF (L2 rel) %.2e | grad ON vs OFF %.2e" % (k, rel(on[k], off[k]), rel(on[k].grad, off[k].grad)))
y = off["box_mlp.0.out"].detach().double()
sd_ = y.std(dim=0)
print("box_mlp.0 output over the 48 clouds: per-channel std min %.3e median %.3e; |mean| median %.3e" % (float(sd_.min()), float(sd_.median()), float(y.mean(0).abs().median())))
x = off["box_mlp.0.in"].detach().double()
d = (on["box_mlp.0.in"].detach().double() - x)
print("X: max abs %.3e, max abs difference ON-OFF %.3e; channels of X with a difference > 1e-5 max: %d" % (float(x.abs().max()), float(d.abs().max()),
      int((d.abs().amax(0) > 1e-5 * float(x.abs().max())).sum())))
gd = on["box_mlp.0.out"].grad.double() - off["box_mlp.0.out"].grad.double()
print("dL/dy0: per-channel relative difference, top 5:", sorted((gd.norm(dim=0) / (off["box_mlp.0.out"].grad.double().norm(dim=0) + 1e-30)).tolist())[-5:])
print("dL/dy0: per-channel std of y0 for the 5 channels with the largest gradient difference:",
      sd_[(gd.norm(dim=0)).topk(5).indices].tolist())
