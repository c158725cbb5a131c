#!/usr/bin/env python
"""GPU diagnostic: which fused launches does one training forward+backward make with / without the prefetched geometry?"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from open3dsot_amd import fused, synth, trackers
name = sys.argv[1] if len(sys.argv) > 1 else "P2B"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = trackers.get_model(name)().to(dev).train()
batch = synth.to_torch(synth.make_batch(100, 48), dev)
for with_geo in (False, True):
    b = dict(batch)
    if with_geo:
        with torch.no_grad():
            b.update(model.sampling_inputs(batch))
    fused._PROF["events"] = []; fused._PROF["on"] = True
    loss, _ = model.training_loss(b); loss.backward(); torch.cuda.synchronize()
    fused._PROF["on"] = False
    c = collections.Counter(e[0] for e in fused._PROF["events"])
    print(name, "geometry given" if with_geo else "inline", "keys", len(b), {k: c[k] for k in ("sample_query", "compact_build", "group_reduce", "conv_fwd")}, "loss %.6f" % float(loss))
