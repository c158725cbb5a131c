"""Diagnostic (GPU box): run-to-run determinism of the fused / composed training step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from open3dsot_amd import sa_modules, synth, trackers, fused as F_
name = sys.argv[1] if len(sys.argv) > 1 else "P2B"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = trackers.get_model(name)().to(dev).train()
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
batch = synth.to_torch(synth.make_batch(300, 3, 256, 512), dev)
def run(fused):
    sa_modules.set_fused(fused)
    model.load_state_dict(sd)
    model.zero_grad(set_to_none=True)
    loss, ld = model.training_loss(batch)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), {k: p.grad.detach().double().cpu() for k, p in model.named_parameters()}
for fused in (True, False):
    runs = [run(fused) for _ in range(6)]
    gmax = max(float(v.norm() / v.numel() ** 0.5) for v in runs[0][1].values())
    keys = [k for k, v in runs[0][1].items() if float(v.norm() / v.numel() ** 0.5) > 1e-3 * gmax]
    print("fused" if fused else "composed", "params judged:", len(keys), "of", len(runs[0][1]))
    for i in range(1, 6):
        rows = sorted(((float((runs[i][1][k] - runs[0][1][k]).norm() / (runs[0][1][k].norm() + 1e-30)), k) for k in keys), reverse=True)
        print("   run %d vs run 0:" % i, "  ".join("%.2e %s" % r for r in rows[:3]))
# unit level: the RPN-shaped SA (xyz and feature gradients live)
import test_fused_gpu as T
for kind in ("rpn", "sa3", "sa2", "sa1"):
    grouper, mlp, xyz, new_xyz, feats = T.make_case(kind, train=True)
    outs = []
    for it in range(6):
        for p in mlp.parameters():
            p.grad = None
        lv = [t.clone().requires_grad_(True) if t is not None else None for t in (xyz, new_xyz, feats)]
        out = F_.sa_group_mlp_pool(grouper, mlp, *lv)
        go = torch.randn(out.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
        out.backward(go)
        torch.cuda.synchronize()
        g = {n: p.grad.detach().double().cpu() for n, p in mlp.named_parameters()}
        for nm, t in zip(("xyz", "new_xyz", "feats"), lv):
            if t is not None and t.grad is not None:
                g[nm] = t.grad.detach().double().cpu()
        g["out"] = out.detach().double().cpu()
        outs.append(g)
    for i in range(1, 6):
        rows = sorted(((float((outs[i][k] - outs[0][k]).norm() / (outs[0][k].norm() + 1e-30)), k) for k in outs[0]), reverse=True)
        print(kind, "iter %d vs 0:" % i, "  ".join("%.2e %s" % r for r in rows[:3]))
