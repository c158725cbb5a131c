"""Diagnostic (GPU box): first divergence between the fused and the composed execution of the
same model on the same batch: per-module forward outputs and output-gradients.  Not a test."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open3dsot_amd import sa_modules, synth, trackers

name = sys.argv[1] if len(sys.argv) > 1 else "BAT"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = trackers.get_model(name)().to(dev).train()
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
batch = synth.to_torch(synth.make_batch(300, 3, 256, 512), dev)
mods = {n: m for n, m in model.named_modules() if n and n.count(".") <= 2 and not n.startswith("backbone.SA_modules.0.") }


def run(fused):
    sa_modules.set_fused(fused)
    model.load_state_dict(sd)
    model.zero_grad(set_to_none=True)
    fw, bw, calls = {}, {}, {}
    hs = []
    def mk(n):
        def hook(mod, inp, out):
            c = calls.get(n, 0); calls[n] = c + 1
            key = "%s#%d" % (n, c)
            outs = out if isinstance(out, (tuple, list)) else (out,)
            for i, o in enumerate(outs):
                if torch.is_tensor(o) and o.dtype.is_floating_point:
                    fw["%s/%d" % (key, i)] = o.detach().double().cpu()
                    if o.requires_grad:
                        o.register_hook(lambda g, k="%s/%d" % (key, i): bw.__setitem__(k, g.detach().double().cpu()))
        return hook
    for n, m in mods.items():
        hs.append(m.register_forward_hook(mk(n)))
    loss, _ = model.training_loss(batch)
    loss.backward()
    torch.cuda.synchronize()
    for h in hs:
        h.remove()
    return fw, bw, {k: p.grad.detach().double().cpu() for k, p in model.named_parameters()}

fc, bc, gc = run(False)
ff, bf, gf = run(True)
def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
def l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))
print("---- forward outputs (fused vs composed), worst first")
rows = sorted(((rel(ff[k], fc[k]), l2(ff[k], fc[k]), k, tuple(fc[k].shape)) for k in fc if k in ff), reverse=True)
for r in rows[:14]:
    print("  %-50s max %.3e l2 %.3e %s" % (r[2], r[0], r[1], r[3]))
print("---- gradients wrt module outputs")
rows = sorted(((rel(bf[k], bc[k]), l2(bf[k], bc[k]), k, tuple(bc[k].shape)) for k in bc if k in bf), reverse=True)
for r in rows[:20]:
    print("  %-50s max %.3e l2 %.3e %s" % (r[2], r[0], r[1], r[3]))
print("---- parameter gradients")
rows = sorted(((rel(gf[k], gc[k]), l2(gf[k], gc[k]), k) for k in gc), reverse=True)
for r in rows[:12]:
    print("  %-55s max %.3e l2 %.3e" % (r[2], r[0], r[1]))
# how many elements of the worst output-gradient differ materially
k = max(((rel(bf[k], bc[k]), k) for k in bc if k in bf))[1]
d = (bf[k] - bc[k]).abs(); s = bc[k].abs().max()
print("worst", k, "elements > 1e-4*scale:", int((d > 1e-4 * s).sum()), "of", d.numel())
