"""Diagnostic (GPU box): where do GPU gradients leave the CPU oracle's?  Not a test."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import torch_ref, ops as O
from open3dsot_amd import sa_modules, synth, trackers, ext

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = trackers.BAT().to(dev).train()
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
for k in sd:
    if sd[k].dtype.is_floating_point and "running" not in k:
        sd[k].requires_grad_(True)
host = synth.make_batch(300, 3, 256, 512)
batch = synth.to_torch(host, dev)


def run(fused):
    sa_modules.set_fused(fused)
    model.load_state_dict({k: v.detach() for k, v in sd.items()})
    model.zero_grad(set_to_none=True)
    ep = model(batch)
    model.load_state_dict({k: v.detach() for k, v in sd.items()})
    loss, ld = model.training_loss(batch)
    loss.backward()
    torch.cuda.synchronize()
    return ep, loss, {k: p.grad.detach().cpu().double() for k, p in model.named_parameters()}

ep_c, loss_c, g_c = run(False)
ep_f, loss_f, g_f = run(True)
torch.backends.cudnn.enabled = False       # hypothesis: MIOpen BatchNorm backward is the outlier
ep_n, loss_n, g_n = run(False)
ep_nf, loss_nf, g_nf = run(True)
torch.backends.cudnn.enabled = True
cpu_batch = synth.to_torch(host)
sd2 = {k: v.detach().clone().requires_grad_(v.requires_grad) for k, v in sd.items()}
out = torch_ref.bat_forward(sd2, cpu_batch, True)
w = {k: v for k, v in vars(model.config).items() if k.endswith("_weight")}
loss_r, _ = torch_ref.matching_loss(cpu_batch, out, w, bat=True)
loss_r.backward()
print("loss gpu composed %.7f fused %.7f cpu %.7f" % (float(loss_c), float(loss_f), float(loss_r)))
for k in ("estimation_boxes", "estimation_cla", "vote_xyz", "center_xyz", "pred_search_bc"):
    a, b = ep_c[k].detach().cpu().double(), out[k].detach().double()
    print("fwd %-18s max|d| %.3e  scale %.3e" % (k, float((a - b).abs().max()), float(b.abs().max())))
print("sample_idxs equal:", bool((ep_c["sample_idxs"].cpu() == out["sample_idxs"]).all()))
# RPN ball query membership GPU vs CPU on their own vote_xyz
vg, vc = ep_c["vote_xyz"].detach(), out["vote_xyz"].detach()
bg = ext.ball_query(vg[:, :64].contiguous(), vg.contiguous(), 0.3, 16).cpu().numpy()
bc = O.ball_query(vc[:, :64].numpy(), vc.numpy(), 0.3, 16)
print("rpn ball idx mismatching groups: %d / %d" % (int((bg != bc).any(-1).sum()), bg.shape[0] * bg.shape[1]))
gmax = max(float(sd2[k].grad.abs().max()) for k in g_c)
rows = []
for k in g_c:
    r = sd2[k].grad.double()
    e_c = float((g_c[k] - r).abs().max()); e_f = float((g_f[k] - r).abs().max()); e_cf = float((g_c[k] - g_f[k]).abs().max())
    rows.append((e_c / (float(r.abs().max()) + 1e-3 * gmax), k, e_c, e_f, e_cf, float(r.abs().max())))
rows.sort(reverse=True)
print("gmax %.3e" % gmax)
worst_n = max(float((g_n[k] - sd2[k].grad.double()).abs().max()) / (float(sd2[k].grad.abs().max()) + 1e-3 * gmax) for k in g_c)
worst_nf = max(float((g_nf[k] - sd2[k].grad.double()).abs().max()) / (float(sd2[k].grad.abs().max()) + 1e-3 * gmax) for k in g_c)
print("WITH cudnn/MIOpen disabled: worst rel composed-vs-cpu %.3e   fused-vs-cpu %.3e" % (worst_n, worst_nf))
print("%-55s %10s %10s %10s %10s %10s" % ("param", "rel(c-cpu)", "|c-cpu|", "|f-cpu|", "|c-f|", "|cpu|max"))
for r in rows[:8]:
    print("%-55s %10.3e %10.3e %10.3e %10.3e %10.3e" % (r[1], r[0], r[2], r[3], r[4], r[5]))
