"""Diagnostic (GPU box): which loss term carries the fused-vs-composed gradient difference."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open3dsot_amd import sa_modules, synth, trackers
name = sys.argv[1] if len(sys.argv) > 1 else "P2B"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
torch.manual_seed(seed)
model = trackers.get_model(name)().to(dev).train()
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
batch = synth.to_torch(synth.make_batch(300 + 17 * seed, 3, 256, 512), dev)
def run(fused, term):
    sa_modules.set_fused(fused)
    model.load_state_dict(sd)
    model.zero_grad(set_to_none=True)
    loss, ld = model.training_loss(batch)
    ld[term].backward()
    torch.cuda.synchronize()
    return float(ld[term].detach()), {k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.grad is not None}
terms = ["loss_objective", "loss_box", "loss_seg", "loss_vote"] + (["loss_bc"] if name == "BAT" else [])
for term in terms:
    lf, gf = run(True, term)
    lc, gc = run(False, term)
    if not gc:
        print(term, "no grad"); continue
    gmax = max(float(v.norm() / v.numel() ** 0.5) for v in gc.values())
    keys = [k for k, v in gc.items() if float(v.norm() / v.numel() ** 0.5) > 1e-3 * gmax]
    rows = sorted(((float((gf[k] - gc[k]).norm() / (gc[k].norm() + 1e-30)), k) for k in keys), reverse=True)
    med = sorted(r[0] for r in rows)[len(rows) // 2] if rows else 0
    tot = sum(float(v.norm() ** 2) for v in gc.values()) ** 0.5
    print("%-15s f %.7f c %.7f |g| %.3e | median l2 %.2e | worst: %s" % (term, lf, lc, tot, med, "  ".join("%.2e %s" % r for r in rows[:2])))
