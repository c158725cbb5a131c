"""Loss terms pinned on outputs of the reference's own code: tests/golden/ref_losses.npz holds what
MatchingBaseModel.compute_loss (models/base_model.py:122-164) and BAT.compute_loss (models/bat.py:57-65) return -- and
the gradients of the weighted total (bat.py:131-137) -- for three seeded cases incl. empty denominators
(tests/golden/make_golden_loss.py).  Checked here: the host mirror (CPU) and the two-launch HIP loss (GPU)."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT_KEYS = ("estimation_cla", "vote_xyz", "estimation_boxes", "center_xyz", "pred_search_bc")
DATA_KEYS = ("seg_label", "box_label", "points2cc_dist_s")
GRAD_KEYS = ("estimation_cla", "vote_xyz", "estimation_boxes", "pred_search_bc")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_losses.npz"))


def case_inputs(gold, c, device="cpu"):
    out = {k: torch.from_numpy(gold["c%d.in.%s" % (c, k)]).to(device).requires_grad_(k in GRAD_KEYS) for k in OUT_KEYS}
    data = {k: torch.from_numpy(gold["c%d.in.%s" % (c, k)]).to(device) for k in DATA_KEYS}
    return data, out


def close(a, b, tol=2e-6):
    a = a.detach() if torch.is_tensor(a) else a
    return abs(float(a) - float(b)) <= tol * (1.0 + abs(float(b)))


@pytest.mark.parametrize("c", range(3))
def test_host_mirror_matches_reference_losses(gold, c):
    from open3dsot_amd import trackers
    data, out = case_inputs(gold, c)
    bat, p2b = trackers.BAT(), trackers.P2B()
    ld = bat.compute_loss(data, out)
    for k in ("loss_objective", "loss_box", "loss_seg", "loss_vote", "loss_bc"):
        assert close(ld[k], gold["c%d.bat.%s" % (c, k)]), (k, float(ld[k]), float(gold["c%d.bat.%s" % (c, k)]))
    for k, v in p2b.compute_loss(data, out).items():
        assert close(v, gold["c%d.p2b.%s" % (c, k)]), k
    cfg = bat.config
    total = (ld["loss_objective"] * cfg.objectiveness_weight + ld["loss_box"] * cfg.box_weight + ld["loss_seg"] * cfg.seg_weight
             + ld["loss_vote"] * cfg.vote_weight + ld["loss_bc"] * cfg.bc_weight)
    assert close(total, gold["c%d.bat.total" % c])
    total.backward()
    for k in GRAD_KEYS:
        ref = gold["c%d.grad.%s" % (c, k)]
        assert np.abs(out[k].grad.numpy() - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-9, k


@pytest.mark.gpu
@pytest.mark.parametrize("c", range(3))
def test_hip_loss_matches_reference_losses(gold, c):
    from open3dsot_amd import fused_loss, trackers
    data, out = case_inputs(gold, c, "cuda")
    total, parts = fused_loss.track_loss(trackers.BAT().config, data, out, True)
    assert close(total, gold["c%d.bat.total" % c], 1e-5), (float(total), float(gold["c%d.bat.total" % c]))
    for k in ("loss_objective", "loss_box", "loss_seg", "loss_vote", "loss_bc"):
        assert close(parts[k], gold["c%d.bat.%s" % (c, k)], 1e-5), k
    total.backward()
    for k in GRAD_KEYS:
        ref = gold["c%d.grad.%s" % (c, k)]
        assert np.abs(out[k].grad.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-8, k
    assert out["center_xyz"].grad is None
