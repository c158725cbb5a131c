"""BoxCloud / regularize_pc: the oracle is pinned against outputs of the
reference's own functions (tests/golden/ref_boxcloud.npz, made by tests/golden/make_golden_boxcloud.py);
the HIP kernel is checked against both."""
import os

import numpy as np
import pytest
import torch

from oracle import boxcloud as obc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_boxcloud.npz"))


@pytest.mark.parametrize("i", range(4))
def test_oracle_matches_reference_boxcloud(gold, i):
    c, s, r, f = gold["center_%d" % i], gold["wlh_%d" % i], gold["rot_%d" % i], float(gold["factor_%d" % i])
    np.testing.assert_allclose(obc.box_corners(c, s, r, f), gold["corners_%d" % i], rtol=0, atol=1e-12)
    np.testing.assert_allclose(obc.get_point_to_box_distance(gold["points_%d" % i], c, s, r, f), gold["bc_%d" % i],
                               rtol=0, atol=1e-12)


@pytest.mark.parametrize("j", range(5))
def test_oracle_matches_reference_regularize(gold, j):
    res, idx = obc.regularize_pc(gold["reg_in_%d" % j], gold["reg_out_%d" % j].shape[0], seed=1)
    assert np.array_equal(res, gold["reg_out_%d" % j])
    want = gold["reg_idx_%d" % j]
    assert (idx is None and want.tolist() == [-1]) or np.array_equal(idx, want)


def test_host_regularize_draws_the_reference_indices(gold):
    from open3dsot_amd import points_utils
    for j in range(5):
        res, idx = points_utils.regularize_pc(torch.from_numpy(gold["reg_in_%d" % j]), gold["reg_out_%d" % j].shape[0], seed=1)
        assert np.array_equal(res.numpy(), gold["reg_out_%d" % j])


def test_boxcloud_fails_loudly_without_gpu():
    from open3dsot_amd import points_utils
    with pytest.raises(RuntimeError):
        points_utils.get_point_to_box_distance(torch.zeros(4, 3), np.zeros(3), np.ones(3), np.eye(3))


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(4))
def test_boxcloud_kernel_matches_reference(gold, i):
    from open3dsot_amd import points_utils
    c, s, r, f = gold["center_%d" % i], gold["wlh_%d" % i], gold["rot_%d" % i], float(gold["factor_%d" % i])
    pts = torch.from_numpy(gold["points_%d" % i]).cuda()
    out = points_utils.get_point_to_box_distance(pts, c, s, r, f).cpu().numpy()
    want = gold["bc_%d" % i]
    assert out.shape == want.shape and out.dtype == np.float32
    # fp32 kernel against the fp64 reference: a few ulp of the coordinates' magnitude
    np.testing.assert_allclose(out, want, rtol=2e-6, atol=2e-6)


@pytest.mark.gpu
def test_boxcloud_kernel_batched_and_properties():
    """batch of boxes at BASELINE sizes: matches the oracle; invariant under a rigid motion of points + box;
    channel 0 is the distance to the centre; ragged tail (N not a multiple of the block) and empty input"""
    from open3dsot_amd import points_utils
    rng = np.random.default_rng(5)
    B, N = 48, 1000
    pts = rng.normal(0, 2, (B, N, 3)).astype(np.float32)
    cen = rng.normal(0, 1, (B, 3)); wlh = rng.uniform(0.5, 4, (B, 3))
    yaw = rng.uniform(-np.pi, np.pi, B)
    rot = np.stack([np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]) for a in yaw])
    out = points_utils.get_point_to_box_distance(torch.from_numpy(pts).cuda(), cen, wlh, rot).cpu().numpy()
    for b in (0, 17, 47):
        np.testing.assert_allclose(out[b], obc.get_point_to_box_distance(pts[b], cen[b], wlh[b], rot[b]), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(out[..., 0], np.linalg.norm(pts - cen[:, None, :].astype(np.float32), axis=-1), rtol=2e-6, atol=2e-6)
    a = 0.7
    Q = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    t = np.array([1.0, -2.0, 0.5])
    out2 = points_utils.get_point_to_box_distance(torch.from_numpy((pts @ Q.T + t).astype(np.float32)).cuda(),
                                                  cen @ Q.T + t, wlh, np.einsum("ij,bjk->bik", Q, rot)).cpu().numpy()
    np.testing.assert_allclose(out2, out, rtol=0, atol=2e-5)
    empty = points_utils.get_point_to_box_distance(torch.zeros(2, 0, 3).cuda(), cen[:2], wlh[:2], rot[:2])
    assert empty.shape == (2, 0, 9)


@pytest.mark.gpu
def test_bat_prepare_input_on_device():
    """models/bat.py:41-55: resample both clouds (seed 1), template BoxCloud -- against the oracle's restatement"""
    from open3dsot_amd import trackers
    rng = np.random.default_rng(9)
    model = trackers.BAT()
    t = rng.normal(0, 1, (300, 3)).astype(np.float32)
    s = rng.normal(0, 2, (1500, 3)).astype(np.float32)
    box = (np.array([0.1, -0.2, 0.3]), np.array([1.6, 3.9, 1.5]), np.eye(3))
    d = model.prepare_input(torch.from_numpy(t).cuda(), torch.from_numpy(s).cuda(), box)
    tp, _ = obc.regularize_pc(t, 512, seed=1)
    sp, _ = obc.regularize_pc(s, 1024, seed=1)
    assert d["template_points"].shape == (1, 512, 3) and d["search_points"].shape == (1, 1024, 3)
    assert np.array_equal(d["template_points"][0].cpu().numpy(), tp) and np.array_equal(d["search_points"][0].cpu().numpy(), sp)
    np.testing.assert_allclose(d["points2cc_dist_t"][0].cpu().numpy(), obc.get_point_to_box_distance(tp, *box), rtol=2e-6, atol=2e-6)
