"""The REFERENCE'S OWN Python operator layer -- pointnet2/utils/pointnet2_utils.py (which does
`import pointnet2_ops._ext as _ext`, :17), pointnet2_modules.py, models/backbone/pointnet.py, models/head/rpn.py --
executed unchanged on the GPU over this repo's drop-in `pointnet2_ops._ext` (ctypes -> libo3dsot_hip.so), and
checked against tests/golden/ref_python_layers.npz (the same layers run over the CPU oracle shim by
tests/golden/make_golden.py).  This is the "LightningModule trackers call it unchanged" claim executed, not
just name-checked (tests/test_capi_symbols.py::test_pointnet2_ops_ext_is_a_dropin).

Needs BOTH a GPU and the reference tree.  /root/reference does not travel to the GPU box (and the build
container has no GPU), so the driver's runs skip it.  Round 3 ran it once on an MI355X from a git-ignored scratch copy
of the nine reference files it imports (tools/ref_scratch.sh; the copy is deleted afterwards and never committed):
    tools/ref_scratch.sh make && gpurun -- 'O3D_REFERENCE_ROOT=$PWD/.refscratch python -m pytest tests/test_reference_modules_gpu.py -m gpu -v'
The log is profiles/r03_reference_modules_over_hip_ext.log.
"""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

REF = os.environ.get("O3D_REFERENCE_ROOT", "/root/reference")
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(os.path.join(REF, "pointnet2", "utils", "pointnet2_utils.py")),
                                 reason="reference tree absent (it cannot travel to the GPU box)"),
              pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")]

TOL = dict(rtol=1e-4, atol=1e-4)


@pytest.fixture(scope="module")
def ref_layers():
    import pointnet2_ops._ext as ext      # this repo's drop-in (repo root is on sys.path: tests/conftest.py)
    assert "open3dsot_amd" in (getattr(ext.furthest_point_sampling, "__module__", "") or "")
    sys.path.insert(0, REF)
    try:
        from pointnet2.utils import pointnet2_modules, pointnet2_utils
        assert pointnet2_utils._ext is ext       # pointnet2_utils.py:17 bound OUR module

        def load(name, rel):
            spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return mod
        yield {"modules": pointnet2_modules, "utils": pointnet2_utils,
               "backbone": load("ref_backbone_gpu", "models/backbone/pointnet.py"),
               "rpn": load("ref_rpn_gpu", "models/head/rpn.py"), "xcorr": load("ref_xcorr_gpu", "models/head/xcorr.py")}
    finally:
        sys.path.remove(REF)


def _sd(golden, prefix):
    return {k[len(prefix):]: torch.from_numpy(golden[k].copy()) for k in golden.files if k.startswith(prefix)}


@pytest.mark.parametrize("use_fps", [True, False])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_reference_backbone_over_hip_ext(golden, ref_layers, use_fps, mode):
    """reference Pointnet_Backbone (models/backbone/pointnet.py:12-88 -> pointnet2_modules.py:31-79 ->
    pointnet2_utils.py:37-339 -> _ext.*) on cuda:0 reproduces the golden outputs"""
    tag = "backbone_fps%d" % int(use_fps)
    net = ref_layers["backbone"].Pointnet_Backbone(use_fps=use_fps, normalize_xyz=False, return_intermediate=True)
    net.load_state_dict(_sd(golden, tag + ".sd."), strict=True)
    net = net.cuda().train(mode == "train")
    pc = torch.from_numpy(golden["search_points"]).cuda()
    N = pc.shape[1]
    xyzs, feats, idx0 = net(pc, [N // 2, N // 4, N // 8])
    assert np.array_equal(idx0.cpu().numpy(), golden["%s.%s.idx0" % (tag, mode)])
    np.testing.assert_allclose(xyzs[-1].detach().cpu().numpy(), golden["%s.%s.xyz" % (tag, mode)], **TOL)
    np.testing.assert_allclose(feats[0].detach().cpu().numpy(), golden["%s.%s.feat0" % (tag, mode)], **TOL)
    np.testing.assert_allclose(feats[-1].detach().cpu().numpy(), golden["%s.%s.feat" % (tag, mode)], **TOL)
    if mode == "train":       # and the backward runs through the reference's autograd Functions (:95-99, :220-239)
        feats[-1].sum().backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_reference_rpn_and_xcorr_over_hip_ext(golden, ref_layers, mode):
    """reference P2BVoteNetRPN (models/head/rpn.py:41-67: SA module with grad-carrying vote_xyz) and BoxAwareXCorr
    (models/head/xcorr.py:67-103: grouping_operation on kNN indices) over the HIP _ext"""
    rpn = ref_layers["rpn"].P2BVoteNetRPN(256, vote_channel=256, num_proposal=64, normalize_xyz=False)
    rpn.load_state_dict(_sd(golden, "rpn.sd."), strict=True)
    rpn = rpn.cuda().train(mode == "train")
    xyz = torch.from_numpy(golden["rpn.in.xyz"]).cuda()
    feat = torch.from_numpy(golden["rpn.in.feat"]).cuda().requires_grad_(True)
    boxes, cla, vote_xyz, centers = rpn(xyz, feat)
    for nm, t in (("boxes", boxes), ("cla", cla), ("vote_xyz", vote_xyz), ("centers", centers)):
        np.testing.assert_allclose(t.detach().cpu().numpy(), golden["rpn.%s.%s" % (mode, nm)], **TOL)
    boxes.sum().backward()          # gather_points_grad + group_points_grad on xyz are live here
    assert torch.isfinite(feat.grad).all()
    gi = {k.split(".")[-1]: torch.from_numpy(golden[k]).cuda() for k in golden.files if k.startswith("xcorr.in.")}
    bax = ref_layers["xcorr"].BoxAwareXCorr(feature_channel=256, hidden_channel=256, out_channel=256, k=4,
                                            use_search_bc=False, use_search_feature=False, bc_channel=9)
    bax.load_state_dict(_sd(golden, "bat_xcorr.sd."), strict=True)
    bax = bax.cuda().train(mode == "train")
    out = bax(gi["t_feat"], gi["s_feat"], gi["t_xyz"], gi["s_xyz"], gi["t_bc"], gi["s_bc"])
    np.testing.assert_allclose(out.detach().cpu().numpy(), golden["bat_xcorr.%s.out" % mode], **TOL)


@pytest.mark.parametrize("name", ["BAT", "P2B"])
def test_reference_tracker_class_trains_over_hip_ext(ref_layers, name):
    """The reference's OWN tracker class -- models/bat.py::BAT / models/p2b.py::P2B on models/base_model.py, Lightning
    & co. stubbed at import level only (tests/golden/ref_stubs.py) -- moved to cuda:0 and run through its own
    `training_step` (forward :82-112, compute_loss, the `.item()` logging) over this repo's `pointnet2_ops._ext`: end
    points, loss and running statistics equal what the same class produced over the CPU oracle shim
    (tests/golden/ref_trackers_b8.npz), and the backward reaches every parameter.  "The LightningModule trackers call it
    unchanged", executed."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import det_init
    import ref_stubs
    from open3dsot_amd import synth, trackers
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_trackers_b8.npz"))
    B, M, N, seed0, wseed = [int(v) for v in gold["meta.shape"]]
    sys.path.insert(0, REF)
    try:
        cls = ref_stubs.load_trackers(REF)[name]
    finally:
        sys.path.remove(REF)
    cfg = ref_stubs.EasyDict(trackers.BAT_CAR if name == "BAT" else trackers.P2B_CAR)
    torch.manual_seed(0)
    model = cls(cfg)
    det_init.fill_state_dict_random(model, seed=wseed)
    model = model.cuda().train()
    batch = synth.to_torch(synth.make_batch(seed0, B, M, N), torch.device("cuda", 0))
    captured = {}
    fwd = model.forward
    model.forward = lambda b: captured.update(fwd(b)) or captured
    loss = model.training_step({k: v.clone() for k, v in batch.items()}, 0)
    loss.backward()
    torch.cuda.synchronize()
    n = 0
    for k in [k for k in gold.files if k.startswith(name + ".train32.")]:
        want, got = gold[k], captured[k.split(".train32.")[1]].detach().cpu().numpy()
        if want.dtype.kind in "iu":
            assert np.array_equal(got, want), k
        else:
            err = float(np.abs(got.astype(np.float64) - want).max() / (np.abs(want).max() + 1e-12))
            assert err < 1e-4, (k, err)
        n += 1
    assert n >= 6
    want = float(gold[name + ".loss32"])
    assert abs(float(loss.detach()) - want) <= 1e-4 * (1 + abs(want)), (float(loss.detach()), want)
    sd = model.state_dict()
    for k in [k for k in gold.files if k.startswith(name + ".after.") and "num_batches" not in k]:
        got, w = sd[k.split(".after.")[1]].cpu().numpy(), gold[k]
        assert float(np.abs(got - w).max() / (np.abs(w).max() + 1e-12)) < 1e-4, k
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    gnorm = float(gold[name + ".gradnorm64"])
    num = den = 0.0
    named = dict(model.named_parameters())
    for k in [k for k in gold.files if k.startswith(name + ".grad64.")]:
        w = gold[k].astype(np.float64).ravel()
        g = named[k.split(".grad64.")[1]].grad.detach().cpu().numpy().astype(np.float64).ravel()
        num += float(((g - w) ** 2).sum()); den += float((w ** 2).sum())
    print("reference %s class over the HIP _ext: stored-key gradient vs the fp64 truth %.2e L2 (the same class over the CPU "
          "oracle shim, all keys: %.2e)" % (name, (num / den) ** 0.5, float(gold[name + ".ref32err_whole"])))
    assert (num / den) ** 0.5 < max(5e-2, 3 * float(gold[name + ".ref32err_whole"])) and gnorm > 0
