"""per-parameter gradient error of the GPU run against the reference classes' golden gradients (tests/golden/ref_trackers.npz)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import det_init
from open3dsot_amd import sa_modules, synth, trackers
gold = np.load(os.path.join(ROOT, "tests", "golden", "ref_trackers.npz"))
dev = torch.device("cuda", 0)
for fused in (True, False):
    sa_modules.set_fused(fused)
    for name in ("BAT", "P2B"):
        model = trackers.get_model(name)(); det_init.fill_state_dict(model); model = model.to(dev).train()
        batch = synth.to_torch(synth.make_batch(40, 2, 256, 512), dev)
        loss, _ = model.training_loss(batch); loss.backward(); torch.cuda.synchronize()
        named = dict(model.named_parameters())
        rows = []
        for k in [k for k in gold.files if k.startswith(name + ".grad.")]:
            g = named[k.split(".grad.")[1]].grad.detach().cpu().numpy().ravel().astype(np.float64)
            w = gold[k].ravel().astype(np.float64)
            rows.append((float(np.linalg.norm(g - w) / (np.linalg.norm(w) + 1e-30)), float(np.linalg.norm(w)), k))
        rows.sort(reverse=True)
        print("fused" if fused else "composed", name, "loss", float(loss), "gold", float(gold[name + ".train.loss"]))
        for r in rows[:8]:
            print("   %.4f  |g|=%.3e  %s" % r)
