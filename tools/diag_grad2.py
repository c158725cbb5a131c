"""Diagnostic (GPU box): GPU gradients (composed / fused, MIOpen on / off) vs the CPU oracle in fp32
and in an fp64 shadow.  Not a test."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import torch_ref
from open3dsot_amd import sa_modules, synth, trackers

name = sys.argv[1] if len(sys.argv) > 1 else "BAT"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = trackers.get_model(name)().to(dev).train()
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
host = synth.make_batch(300, 3, 256, 512)
batch = synth.to_torch(host, dev)


def run(fused):
    sa_modules.set_fused(fused)
    model.load_state_dict(sd)
    model.zero_grad(set_to_none=True)
    loss, ld = model.training_loss(batch)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.detach().cpu().double() for k, p in model.named_parameters()}


def cpu(dtype):
    sd2 = {k: (v.detach().clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    for k, v in sd2.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    b = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in synth.to_torch(host).items()}
    fwd = torch_ref.bat_forward if name == "BAT" else torch_ref.p2b_forward
    out = fwd(sd2, b, True)
    w = {k: v for k, v in vars(model.config).items() if k.endswith("_weight")}
    loss, _ = torch_ref.matching_loss(b, out, w, bat=name == "BAT")
    loss.backward()
    return float(loss), {k: sd2[k].grad.double() for k, _ in model.named_parameters()}

res = {}
res["gpu_composed"] = run(False)
res["gpu_fused"] = run(True)
torch.backends.cudnn.enabled = False
res["gpu_composed_nomiopen"] = run(False)
res["gpu_fused_nomiopen"] = run(True)
torch.backends.cudnn.enabled = True
res["cpu32"] = cpu(torch.float32)
res["cpu64"] = cpu(torch.float64)
for k, (l, _) in res.items():
    print("%-24s loss %.9f" % (k, l))
truth = res["cpu64"][1]
gmax = max(float(v.abs().max()) for v in truth.values())
print("gmax %.3e" % gmax)
for k, (_, g) in res.items():
    if k == "cpu64":
        continue
    rows = sorted(((float((g[p] - truth[p]).abs().max()) / (float(truth[p].abs().max()) + 1e-3 * gmax),
                    float((g[p] - truth[p]).norm() / (truth[p].norm() + 1e-12)), p) for p in truth), reverse=True)
    print("== %s vs cpu64: worst max-rel %.3e" % (k, rows[0][0]))
    for r in rows[:4]:
        print("     %-55s maxrel %.3e  l2rel %.3e" % (r[2], r[0], r[1]))
    for r in rows:
        if r[2] == "backbone.SA_modules.0.mlps.0.layer0.conv.weight":
            print("     %-55s maxrel %.3e  l2rel %.3e  |truth|max %.3e" % (r[2], r[0], r[1], float(truth[r[2]].abs().max())))
    if k == "gpu_fused":
        pn = "backbone.SA_modules.0.mlps.0.layer0.conv.weight"
        print("fused  ", g[pn][:3].flatten().tolist()); print("truth  ", truth[pn][:3].flatten().tolist())
        print("compos ", res["gpu_composed"][1][pn][:3].flatten().tolist())
