"""Generate tests/golden/*.npz by running the REFERENCE'S OWN Python layers in this container.

Run from the repo root, only where /root/reference exists (the build container):
    python tests/golden/make_golden.py
What is reference code here: pointnet2/utils/{pytorch_utils,pointnet2_utils,pointnet2_modules}.py,
models/backbone/pointnet.py, models/head/{xcorr,rpn}.py -- imported read-only from
/root/reference.  What is NOT: `pointnet2_ops._ext` (un-vendored, CUDA-only) is replaced
by oracle/ext_shim.py (the C restatement), and `Tensor.cuda()` is patched to the identity
because the reference hard-codes `.cuda()` at pointnet2_modules.py:56.  The fixtures
therefore pin the Python-layer composition (grouping order, conv/bn/relu order, bias
rule, max-pool axis, head wiring) against the reference itself; the index arithmetic
stays pinned only by the oracle (PARITY UNPINNED upstream, see oracle/pointnet2_oracle.c).
BoxAwareXCorr note: the reference selects neighbours with cdist+argsort (unspecified tie
order); inputs here are continuous random BoxClouds, so no ties occur.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import ext_shim  # noqa: E402

ext_shim.install()
torch.Tensor.cuda = lambda self, *a, **k: self  # reference hard-codes .cuda()
sys.path.insert(0, REF)
from pointnet2.utils import pointnet2_modules as ref_modules  # noqa: E402,F401


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_backbone = _load("ref_backbone", "models/backbone/pointnet.py")
ref_xcorr = _load("ref_xcorr", "models/head/xcorr.py")
ref_rpn = _load("ref_rpn", "models/head/rpn.py")

from open3dsot_amd import synth  # noqa: E402  (numpy-only data generator)

OUT = os.path.dirname(os.path.abspath(__file__))


def sd_np(module, prefix=""):
    return {prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def randomize_bn(module, g):
    """non-trivial BN affine + running stats so eval-mode fixtures exercise them"""
    for m in module.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.weight.data = torch.empty_like(m.weight).uniform_(0.5, 1.5, generator=g)
            m.bias.data = torch.empty_like(m.bias).normal_(0, 0.1, generator=g)
            m.running_mean.data = torch.empty_like(m.running_mean).normal_(0, 0.1, generator=g)
            m.running_var.data = torch.empty_like(m.running_var).uniform_(0.5, 1.5, generator=g)


def main():
    g = torch.Generator().manual_seed(20240925)
    torch.manual_seed(20240925)
    fix = {}
    batch = synth.make_batch(0, 2, template_size=128, search_size=256)
    search = torch.from_numpy(batch["search_points"])

    # ---- backbone (3 SA levels), FPS and prefix variants, train and eval mode ----------
    for use_fps in (True, False):
        net = ref_backbone.Pointnet_Backbone(use_fps=use_fps, normalize_xyz=False, return_intermediate=True)
        randomize_bn(net, g)
        tag = "backbone_fps%d" % int(use_fps)
        for k, v in sd_np(net).items():
            fix["%s.sd.%s" % (tag, k)] = v
        N = search.shape[1]
        for mode in ("train", "eval"):
            net.train(mode == "train")
            # a fresh copy so train-mode running-stat updates do not leak into the eval fixture
            import copy
            n2 = copy.deepcopy(net)
            xyzs, feats, idx0 = n2(search, [N // 2, N // 4, N // 8])
            fix["%s.%s.xyz" % (tag, mode)] = xyzs[-1].detach().numpy()
            fix["%s.%s.feat" % (tag, mode)] = feats[-1].detach().numpy()
            fix["%s.%s.feat0" % (tag, mode)] = feats[0].detach().numpy()
            fix["%s.%s.idx0" % (tag, mode)] = idx0.numpy()
            if mode == "train":
                for k, v in sd_np(n2).items():
                    if "running" in k or "num_batches" in k:
                        fix["%s.train.sd_after.%s" % (tag, k)] = v
    fix["search_points"] = batch["search_points"]

    # ---- BoxAwareXCorr + P2B_XCorr ------------------------------------------------------
    # the trackers' own shapes (64 template / 128 search seeds) and four pairs: a BatchNorm channel of `fea_layer` sees 512
    # samples (round 2's 2 x 32 made the train-mode output ill-conditioned: 1e-2 on the GPU)
    B, f, M, N = 4, 256, 64, 128
    t_feat = torch.randn(B, f, M, generator=g)
    s_feat = torch.randn(B, f, N, generator=g)
    t_xyz = torch.randn(B, M, 3, generator=g)
    s_xyz = torch.randn(B, N, 3, generator=g)
    t_bc = torch.rand(B, M, 9, generator=g) * 3
    s_bc = torch.rand(B, N, 9, generator=g) * 3
    for name, arr in (("t_feat", t_feat), ("s_feat", s_feat), ("t_xyz", t_xyz), ("s_xyz", s_xyz),
                      ("t_bc", t_bc), ("s_bc", s_bc)):
        fix["xcorr.in." + name] = arr.numpy()
    bax = ref_xcorr.BoxAwareXCorr(feature_channel=f, hidden_channel=256, out_channel=256, k=4,
                                  use_search_bc=False, use_search_feature=False, bc_channel=9)
    randomize_bn(bax, g)
    for k, v in sd_np(bax).items():
        fix["bat_xcorr.sd." + k] = v
    for mode in ("train", "eval"):
        import copy
        m2 = copy.deepcopy(bax).train(mode == "train")
        fix["bat_xcorr.%s.out" % mode] = m2(t_feat, s_feat, t_xyz, s_xyz, t_bc, s_bc).detach().numpy()
    pxc = ref_xcorr.P2B_XCorr(feature_channel=f, hidden_channel=256, out_channel=256)
    randomize_bn(pxc, g)
    for k, v in sd_np(pxc).items():
        fix["p2b_xcorr.sd." + k] = v
    for mode in ("train", "eval"):
        import copy
        m2 = copy.deepcopy(pxc).train(mode == "train")
        fix["p2b_xcorr.%s.out" % mode] = m2(t_feat, s_feat, t_xyz).detach().numpy()

    # ---- RPN -----------------------------------------------------------------------------
    N = 128
    batch4 = synth.make_batch(10, 4, template_size=128, search_size=256)
    xyz = torch.from_numpy(batch4["search_points"][:, :N, :]).clone() * 0.3
    feat = torch.randn(4, 256, N, generator=g)
    fix["rpn.in.xyz"], fix["rpn.in.feat"] = xyz.numpy(), feat.numpy()
    rpn = ref_rpn.P2BVoteNetRPN(256, vote_channel=256, num_proposal=64, normalize_xyz=False)
    randomize_bn(rpn, g)
    # make the vote offsets small so the vote ball queries are non-trivial
    with torch.no_grad():
        rpn.vote_layer[2].conv.weight.mul_(0.05)
    for k, v in sd_np(rpn).items():
        fix["rpn.sd." + k] = v
    for mode in ("train", "eval"):
        import copy
        m2 = copy.deepcopy(rpn).train(mode == "train")
        boxes, cla, vote_xyz, centers = m2(xyz, feat)
        for nm, t in (("boxes", boxes), ("cla", cla), ("vote_xyz", vote_xyz), ("centers", centers)):
            fix["rpn.%s.%s" % (mode, nm)] = t.detach().numpy()

    path = os.path.join(OUT, "ref_python_layers.npz")
    np.savez_compressed(path, **{k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in fix.items()})
    print("wrote", path, "%.1f MB" % (os.path.getsize(path) / 1e6), len(fix), "arrays")


if __name__ == "__main__":
    main()
