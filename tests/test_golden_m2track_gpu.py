"""M2-Track on the GPU (flat-GEMM per-point stacks on the library's kernels, restructuring 5) against the reference's
own model: tests/golden/ref_m2track.npz holds what /root/reference/models/m2track.py produced (forward, compute_loss,
BatchNorm running statistics; tests/golden/make_golden_m2track.py).  GPU twin of tests/test_golden_m2track.py."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_m2track.npz"))


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_gpu_m2track_matches_reference(gold, mode):
    from open3dsot_amd import m2track, nn_blocks
    assert nn_blocks._FLAT["on"]
    net = m2track.M2TRACK()
    net.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd.")}, strict=True)
    net = net.cuda().train(mode == "train")
    b = {k[3:]: torch.from_numpy(gold[k]).cuda() for k in gold.files if k.startswith("in.")}
    with torch.set_grad_enabled(mode == "train"):
        out = net(b)
        ld = net.compute_loss(b, out)
    # the heads run BatchNorm1d over the fixture's batch of 8 clouds: a 1e-7 change of the pooled features (the GEMM
    # kernels sum in their own order) moves a normalised value by 1e-4 -- the bound of the CPU twin's flat path
    for k in out:
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), gold["%s.out.%s" % (mode, k)], err_msg=k, rtol=2e-3, atol=5e-4)
    for k in ld:
        assert abs(float(ld[k]) - float(gold["%s.loss.%s" % (mode, k)])) < 1e-3 * (1 + abs(float(ld[k]))), k
    if mode == "train":
        for k, v in net.state_dict().items():
            if "running" in k:
                np.testing.assert_allclose(v.cpu().numpy(), gold["train.sd_after." + k], rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.fixture(scope="module")
def gold48():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_m2track_b48.npz"))


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_gpu_m2track_matches_reference_at_batch48_losses_1e4(gold, gold48, mode):
    """the reference's own M2TRACK on the benchmarked batch of 48 frame pairs (tests/golden/make_golden_m2track.py, second
    fixture): every loss term of the GPU run within 1e-4 (round 3 held 1e-3 on the 8-cloud fixture, whose BatchNorm1d rows
    over 8 samples amplify 1e-7 to 1e-4), end points 1e-3, running statistics 1e-4"""
    from open3dsot_amd import m2track
    net = m2track.M2TRACK()
    net.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd.")}, strict=True)
    net = net.cuda().train(mode == "train")
    b = {k[3:]: torch.from_numpy(gold48[k]).cuda() for k in gold48.files if k.startswith("in.")}
    with torch.set_grad_enabled(mode == "train"):
        out = net(b)
        ld = net.compute_loss(b, out)
    for k in ld:
        want = float(gold48["%s.loss.%s" % (mode, k)])
        assert abs(float(ld[k]) - want) <= 1e-4 * (1 + abs(want)), (k, float(ld[k]), want)
    for k in ("estimation_boxes", "motion_cls", "estimation_boxes_prev"):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), gold48["%s.out.%s" % (mode, k)], err_msg=k, rtol=1e-3, atol=2e-4)
    if mode == "train":
        for k, v in net.state_dict().items():
            if "running" in k:
                np.testing.assert_allclose(v.cpu().numpy(), gold48["train.sd_after." + k], rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("name", ["plain", "no_bc", "no_cls_no_prev", "one_stage"])
def test_gpu_m2track_flag_variants_match_the_reference_class(name, mode):
    """the reference's own M2TRACK with its configuration flags switched (tests/golden/ref_m2track_flags.npz, generator
    tests/golden/make_golden_m2track_flags.py): the GPU path -- per-point stacks, row kernels, motion_merge with / without
    the previous-box refinement, the loss operator with terms switched off -- gives the same output keys, outputs and loss
    terms (batch 8: the BatchNorm1d rows' amplification bounds of the default-configuration test), and its backward runs"""
    from test_golden_m2track import FLAG_VARIANTS, build_variant
    gold = np.load(os.path.join(ROOT, "tests", "golden", "ref_m2track_flags.npz"))
    net = build_variant(name).cuda().train(mode == "train")
    b = {k[3:]: torch.from_numpy(gold[k]).cuda() for k in gold.files if k.startswith("in.")}
    with torch.set_grad_enabled(mode == "train"):
        out = net(b)
        ld = net.compute_loss(b, out)
    pre = "%s.%s." % (name, mode)
    assert set(out) == {k[len(pre) + 4:] for k in gold.files if k.startswith(pre + "out.")}
    assert set(ld) == {k[len(pre) + 5:] for k in gold.files if k.startswith(pre + "loss.")}
    # hard masks (argmax of the segmentation / motion-state logits) gate everything downstream: a flip is only acceptable
    # where the reference's own two logits are within rounding of a tie, and then nothing behind the mask is comparable
    flips = 0
    for key in ("seg_logits", "motion_cls"):
        if key not in out:
            continue
        ref = gold[pre + "out." + key]
        np.testing.assert_allclose(out[key].detach().cpu().numpy(), ref, err_msg=key, rtol=2e-3, atol=5e-4)
        differ = out[key].argmax(1).cpu().numpy() != ref.argmax(1)
        if differ.any():
            margin = np.abs(ref[:, 0] - ref[:, 1])[differ]
            assert float(margin.max()) < 2e-3, (name, mode, key, int(differ.sum()), float(margin.max()))
            flips += int(differ.sum())
        if flips:         # (the motion-state logits sit behind the segmentation mask themselves)
            pytest.skip("%d hard-mask decision(s) of %s within 2e-3 of a tie flipped: outputs behind the mask are not comparable"
                        % (flips, key))
    for k in out:
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), gold[pre + "out." + k], err_msg=k, rtol=2e-3, atol=5e-4)
    for k in ld:
        assert abs(float(ld[k]) - float(gold[pre + "loss." + k])) < 1e-3 * (1 + abs(float(ld[k]))), (k, float(ld[k]))
    if mode == "train":
        ld["loss_total"].backward()
        used = FLAG_VARIANTS[name]
        for pname, p in net.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), pname
        assert used["use_second_stage"] == any(n.startswith("box_mlp") for n, _ in net.named_parameters())
