"""M2-Track against the reference's own code: tests/golden/ref_m2track.npz is
produced by tests/golden/make_golden_m2track.py, which runs /root/reference/models/m2track.py (forward and
compute_loss), models/backbone/pointnet.py and the tensor helpers of datasets/points_utils.py.  The
product's mirror (open3dsot_amd/m2track.py, backbone.py, box_utils.py) loads the reference state_dict
with strict=True and must reproduce outputs, losses and BatchNorm running statistics, on the module-by-
module path and on the flat-GEMM path of the per-point stacks."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = dict(rtol=2e-4, atol=2e-5)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_m2track.npz"))


def build(gold, train):
    from open3dsot_amd import m2track
    net = m2track.M2TRACK()
    sd = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd.")}
    net.load_state_dict(sd, strict=True)
    return net.train(train)


def batch(gold):
    return {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("in.")}


@pytest.mark.parametrize("flat", [True, False])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_m2track_matches_reference(gold, mode, flat):
    from open3dsot_amd import nn_blocks
    was = nn_blocks._FLAT["on"]
    nn_blocks.set_flat_pointwise(flat)
    try:
        net = build(gold, mode == "train")
        b = batch(gold)
        out = net(b)
        ld = net.compute_loss(b, out)
    finally:
        nn_blocks.set_flat_pointwise(was)
    # the heads run BatchNorm1d over a batch of 8 samples: a 1e-7 change of the pooled features (the flat
    # path sums in GEMM order) moves a normalised value by 1e-4; the module-by-module path is bitwise
    # the reference's arithmetic and keeps the tight tolerance
    tol = dict(rtol=2e-3, atol=5e-4) if flat else TOL
    for k in out:
        np.testing.assert_allclose(out[k].detach().numpy(), gold["%s.out.%s" % (mode, k)], err_msg=k, **tol)
    for k in ld:
        assert abs(float(ld[k]) - float(gold["%s.loss.%s" % (mode, k)])) < (1e-3 if flat else 2e-4) * (1 + abs(float(ld[k]))), k
    if mode == "train":
        for k, v in net.state_dict().items():
            if "running" in k:
                np.testing.assert_allclose(v.numpy(), gold["train.sd_after." + k], rtol=1e-4, atol=1e-5, err_msg=k)


def test_box_helpers_match_reference(gold):
    from open3dsot_amd import box_utils
    pts, ref, off = (torch.from_numpy(gold["box.in." + k]) for k in ("pts", "ref", "off"))
    keep = pts.clone()
    np.testing.assert_allclose(box_utils.get_offset_points_tensor(pts, ref, off).numpy(), gold["box.offset_points"], **TOL)
    np.testing.assert_allclose(box_utils.get_offset_box_tensor(ref, off).numpy(), gold["box.offset_box"], **TOL)
    np.testing.assert_allclose(box_utils.remove_transform_points_tensor(pts, ref).numpy(), gold["box.remove_transform"], **TOL)
    assert torch.equal(pts, keep)        # unlike the reference, the inputs are not modified in place


def test_m2track_backward_and_synthetic_contract():
    from open3dsot_amd import m2track, synth
    torch.manual_seed(0)
    net = m2track.M2TRACK().train()
    b = synth.to_torch(synth.make_motion_batch(0, 3, 64))
    assert b["points"].shape == (3, 128, 5) and b["candidate_bc"].shape == (3, 128, 9) and b["seg_label"].dtype == torch.int64
    loss, ld = net.training_loss(b)
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
