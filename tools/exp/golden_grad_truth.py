"""Which of {fused GPU fp32, composed GPU fp32 (torch ops), the reference's CPU fp32 golden} is closest to an fp64
evaluation of the same training step?  (batch of TWO pairs, tests/golden/ref_trackers.npz.)  The fp64 run uses the
composed path with the index operators fed float32 copies (indices are exact) and the gathers written with torch."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import det_init
import open3dsot_amd.ext as ext
from open3dsot_amd import sa_modules, synth, trackers
gold = np.load(os.path.join(ROOT, "tests", "golden", "ref_trackers.npz"))
dev = torch.device("cuda", 0)
real = {n: getattr(ext, n) for n in ("furthest_point_sampling", "furthest_point_sampling_pair", "ball_query", "knn",
                                     "group_points", "group_points_grad", "gather_points", "gather_points_grad", "gather_rows")}


def patch64():
    ext.furthest_point_sampling = lambda x, n: real["furthest_point_sampling"](x.float().contiguous(), n)
    ext.furthest_point_sampling_pair = lambda a, na, b, nb: real["furthest_point_sampling_pair"](a.float().contiguous(), na, b.float().contiguous(), nb)
    ext.ball_query = lambda c, x, r, n: real["ball_query"](c.float().contiguous(), x.float().contiguous(), r, n)
    ext.knn = lambda q, r, k: real["knn"](q.float().contiguous(), r.float().contiguous(), k)
    ext.group_points = lambda f, i: torch.gather(f.unsqueeze(2).expand(-1, -1, i.shape[1], -1), 3, i.long().unsqueeze(1).expand(-1, f.shape[1], -1, -1))
    ext.group_points_grad = lambda g, i, n: torch.zeros(g.shape[0], g.shape[1], n, dtype=g.dtype, device=g.device).scatter_add_(
        2, i.reshape(i.shape[0], 1, -1).expand(-1, g.shape[1], -1).long(), g.reshape(g.shape[0], g.shape[1], -1))
    ext.gather_points = lambda f, i: torch.gather(f, 2, i.long().unsqueeze(1).expand(-1, f.shape[1], -1))
    ext.gather_points_grad = lambda g, i, n: torch.zeros(g.shape[0], g.shape[1], n, dtype=g.dtype, device=g.device).scatter_add_(
        2, i.long().unsqueeze(1).expand(-1, g.shape[1], -1), g)
    ext.gather_rows = lambda s, i: torch.gather(s, 1, i.long().unsqueeze(-1).expand(-1, -1, s.shape[2]))


def unpatch():
    for n, f in real.items():
        setattr(ext, n, f)


def grads(name, fused, dbl):
    sa_modules.set_fused(fused)
    model = trackers.get_model(name)(); det_init.fill_state_dict(model); model = model.to(dev).train()
    batch = synth.to_torch(synth.make_batch(40, 2, 256, 512), dev)
    if dbl:
        model = model.double()
        batch = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
        patch64()
    try:
        loss, _ = model.training_loss(batch); loss.backward(); torch.cuda.synchronize()
    finally:
        unpatch()
    return float(loss.detach()), {k: p.grad.detach().double().cpu().numpy().ravel() for k, p in model.named_parameters() if p.grad is not None}


for name in ("BAT", "P2B"):
    lt, truth = grads(name, False, True)
    lf, fused = grads(name, True, False)
    lc, comp = grads(name, False, False)
    print(name, "loss fp64 %.7f fused %.7f composed %.7f reference-CPU %.7f" % (lt, lf, lc, float(gold[name + ".train.loss"])))
    def e(a, k): return float(np.linalg.norm(a - truth[k]) / (np.linalg.norm(truth[k]) + 1e-30))
    tot = lambda d: float(np.sqrt(sum(np.sum((d[k] - truth[k]) ** 2) for k in truth)) / np.sqrt(sum(np.sum(truth[k] ** 2) for k in truth)))
    print("   whole gradient, relative L2 error against fp64: fused %.4f  composed %.4f" % (tot(fused), tot(comp)))
    for k in [k for k in gold.files if k.startswith(name + ".grad.")]:
        key = k.split(".grad.")[1]
        print("   %-55s |g|=%.2e  fused %.4f  composed %.4f  reference-CPU %.4f" % (
            key, float(np.linalg.norm(truth[key])), e(fused[key], key), e(comp[key], key), e(gold[k].ravel().astype(np.float64), key)))
    worst = sorted(((e(fused[k], k), e(comp[k], k), float(np.linalg.norm(truth[k])), k) for k in truth), reverse=True)[:6]
    for w in worst:
        print("   worst fused: %.4f (composed %.4f) |g|=%.2e %s" % w)
