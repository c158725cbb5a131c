"""Diagnostic (GPU box): fused vs composed parameter gradients over several batches / batch sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open3dsot_amd import sa_modules, synth, trackers
name = sys.argv[1] if len(sys.argv) > 1 else "P2B"
dev = torch.device("cuda", 0)
for B, M, N in ((3, 256, 512), (8, 512, 1024)):
    for seed in range(4):
        torch.manual_seed(seed)
        model = trackers.get_model(name)().to(dev).train()
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        batch = synth.to_torch(synth.make_batch(300 + 17 * seed, B, M, N), dev)
        def run(fused):
            sa_modules.set_fused(fused)
            model.load_state_dict(sd)
            model.zero_grad(set_to_none=True)
            loss, ld = model.training_loss(batch)
            loss.backward()
            torch.cuda.synchronize()
            return float(loss.detach()), {k: p.grad.detach().double().cpu() for k, p in model.named_parameters()}
        lf, gf = run(True)
        lc, gc = run(False)
        gmax = max(float(v.norm() / v.numel() ** 0.5) for v in gc.values())
        keys = [k for k, v in gc.items() if float(v.norm() / v.numel() ** 0.5) > 1e-3 * gmax]
        rows = sorted(((float((gf[k] - gc[k]).norm() / (gc[k].norm() + 1e-30)), k) for k in keys), reverse=True)
        med = sorted(r[0] for r in rows)[len(rows) // 2]
        print("%s B=%d seed %d loss f %.7f c %.7f | median l2 %.2e | worst: %s" % (name, B, seed, lf, lc, med, "  ".join("%.2e %s" % r for r in rows[:3])))
