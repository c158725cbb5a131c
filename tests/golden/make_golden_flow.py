"""Generate tests/golden/ref_flow_modules.npz from the REFERENCE'S OWN `FlowEmbedding` and `PointNetSetUpConv`
(pointnet2/utils/pointnet2_modules.py:215-334; SURVEY.md section 1-L1 lists them in the operator API, no tracker uses them).

Run from the repo root, only where /root/reference exists (the build container):
    python tests/golden/make_golden_flow.py
`pointnet2_ops._ext` is the oracle shim (oracle/ext_shim.py), as in make_golden_cold.py.  Both modules run with knn=True,
the only mode in which the reference's code executes at all (with knn=False it unpacks `idx, cnt` from a ball_query that
returns one tensor, :254 / :311); inputs are continuous random clouds, so the cdist + argsort neighbour selection has no
ties.  Recorded: state dict, inputs, forward output (train and eval), both feature gradients, running statistics.
"""
import copy
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import ext_shim  # noqa: E402

ext_shim.install()
sys.path.insert(0, REF)
warnings.simplefilter("ignore", SyntaxWarning)        # `is 'concat'` / `is not 0` in the reference
from pointnet2.utils import pointnet2_modules as ref_modules  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def randomise(mod, g):
    for m in mod.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.weight.data.uniform_(0.5, 1.5, generator=g)
            m.bias.data.normal_(0, 0.1, generator=g)
            m.running_mean.data.normal_(0, 0.1, generator=g)
            m.running_var.data.uniform_(0.5, 1.5, generator=g)


def record(fix, tag, mod, inputs, grad_idx):
    for k, v in mod.state_dict().items():
        fix["%s.sd.%s" % (tag, k)] = v.detach().numpy().copy()
    for i, t in enumerate(inputs):
        if t is not None:
            fix["%s.in.%d" % (tag, i)] = t.numpy()
    for mode in ("train", "eval"):
        m = copy.deepcopy(mod).train(mode == "train")
        args = [t.clone().requires_grad_(i in grad_idx) if t is not None else None for i, t in enumerate(inputs)]
        out = m(*args)
        out = out[1] if isinstance(out, tuple) else out
        ct = torch.randn(out.shape, generator=torch.Generator().manual_seed(5))
        (out * ct).sum().backward()
        fix["%s.%s.out" % (tag, mode)] = out.detach().numpy()
        for i in grad_idx:
            fix["%s.%s.grad.%d" % (tag, mode, i)] = args[i].grad.numpy()
        if mode == "train":
            fix["%s.ct" % tag] = ct.numpy()
            for k, v in m.state_dict().items():
                if "running" in k:
                    fix["%s.train.sd_after.%s" % (tag, k)] = v.detach().numpy().copy()


def main():
    g = torch.Generator().manual_seed(4242)
    torch.manual_seed(4242)
    fix = {}
    B, N, C = 2, 48, 16
    xyz1, xyz2 = torch.randn(B, N, 3, generator=g), torch.randn(B, N, 3, generator=g)
    f1, f2 = torch.randn(B, C, N, generator=g), torch.randn(B, C, N, generator=g)
    fe = ref_modules.FlowEmbedding(radius=1.0, nsample=8, in_channel=C, mlp=[32, 24], knn=True)
    randomise(fe, g)
    record(fix, "fe", fe, [xyz1, xyz2, f1, f2], (2, 3))
    N1, N2, C1, C2 = 40, 16, 12, 20
    a, b = torch.randn(B, N1, 3, generator=g), torch.randn(B, N2, 3, generator=g)
    fa, fb = torch.randn(B, C1, N1, generator=g), torch.randn(B, C2, N2, generator=g)
    up = ref_modules.PointNetSetUpConv(nsample=4, radius=1.0, f1_channel=C1, f2_channel=C2, mlp=[24, 16], mlp2=[20], knn=True)
    randomise(up, g)
    record(fix, "up", up, [a, b, fa, fb], (2, 3))
    path = os.path.join(OUT, "ref_flow_modules.npz")
    np.savez_compressed(path, **fix)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1e3), len(fix), "arrays")


if __name__ == "__main__":
    main()
