#!/usr/bin/env python
"""GPU diagnostic: per-key norm ratio / cosine of the M2-Track gradient at 48 x 2 048 vs the reference's fp64, cloud bias on/off,
and the gradient of the three heads' inputs"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import test_golden_m2track as T
from open3dsot_amd import backbone, m2track

gold = np.load(os.path.join(ROOT, "tests/golden/ref_m2track.npz"))
gold48 = np.load(os.path.join(ROOT, "tests/golden/ref_m2track_b48.npz"))
goldg = np.load(os.path.join(ROOT, "tests/golden/ref_m2track_grad.npz"))
tag = "b48x2048"
keys = ["box_mlp.6.weight", "box_mlp.6.bias", "box_mlp.4.weight", "box_mlp.3.weight", "box_mlp.1.weight", "box_mlp.0.weight",
        "mini_pointnet2.features.17.weight", "mini_pointnet2.features.14.weight", "mini_pointnet2.features.9.weight",
        "final_mlp.6.weight", "final_mlp.0.weight", "motion_mlp.6.weight", "motion_state_mlp.6.weight", "seg_pointnet.fc.weight"]
for cb in (True, False):
    backbone.set_cloud_bias(cb)
    net = m2track.M2TRACK()
    net.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd.")}, strict=True)
    net = net.cuda().train()
    b = {k: v.cuda() for k, v in T.grad_fixture_batch(tag, gold, gold48, goldg).items()}
    with T.replay_hard_masks(goldg, tag):
        out = net(b)
        ld = net.compute_loss(b, out)
    for k in ("loss_center", "loss_angle", "loss_center_aux", "loss_angle_aux", "loss_seg", "loss_bc"):
        print("   %s %.7f" % (k, float(ld[k].detach())), end="")
    print()
    ld["loss_total"].backward()
    g = dict(net.named_parameters())
    print("cloud bias", cb)
    for k in keys:
        stride = int(goldg["%s.stride.%s" % (tag, k)])
        t = goldg["%s.grad64.%s" % (tag, k)].astype(np.float64)
        a = g[k].grad.detach().double().cpu().flatten()[::stride].numpy()
        print("   %-40s err %.2e  norm ratio %.5f  cos %.6f" % (k, np.linalg.norm(a - t) / np.linalg.norm(t), np.linalg.norm(a) / np.linalg.norm(t),
                                                                  float(a @ t / (np.linalg.norm(a) * np.linalg.norm(t)))))
backbone.set_cloud_bias(True)
for k in ("loss_center", "loss_angle", "loss_center_aux", "loss_angle_aux", "loss_seg", "loss_bc"):
    pass
